"""Diagnostic (GPU box): the rollout parity test's comparison (HIP rollout == composed oracle rollout, every decision and count)
over longer runs and more seeds than the test suite affords.   python tools/diag/parity_long.py [steps] [seeds]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_rollout_parity as T
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for k in range(first, first + seeds):
    t0 = time.time()
    tmp = tempfile.mkdtemp()
    hip_ro, ora, mesh = T._both_rollouts(tmp, cells=8, size=4.8, tess=0.3, scene_seed=k, seed=5 + k)
    mag = []        # steps beyond the magnitude bound are recorded, the run goes on: the DECISIONS are what must hold (asserted hard)
    try:
        counts = T._step_both_and_compare(hip_ro, ora, steps, magnitude_log=mag)
    except AssertionError as e:        # a decision, a count or the statistics over the steps
        print(f"scene {k}: FAILED {str(e)[:600]}", flush=True)
        continue
    for (s_, B_, emax, emean, cmax, cmean, rng) in mag:
        print(f"scene {k}: step {s_} B = {B_}: max |HIP - fp64| {emax:.4g} (mean {emean:.3g}) = {emax / rng:.2e} x range {rng:.1f}; torch-CPU fp32 "
              f"{cmax:.4g} (mean {cmean:.3g}) = {cmax / rng:.2e} x range; ratio {emax / max(cmax, 1e-30):.1f} -- beyond the magnitude bound, "
              f"decisions identical", flush=True)
    print(f"scene {k}: {steps} steps identical (poses, replans {hip_ro.n_replans}, collision / passable lists, cloud, maps, network inputs, "
          f"coverage counts {counts[0]} -> {counts[-1]}), native search {hip_ro.planner.native_search}, {time.time() - t0:.0f} s", flush=True)
