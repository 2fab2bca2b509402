"""Exploration rollout -- host-side mirror of next_best_path/testers/nbp_planning.py
(compute_nbp_trajectory :23-361, test_nbp_planning :364-516): same entry-point names, same
step order (S1-S14 of SURVEY.md section 3.1), same result-JSON schema; every tensor op of the
step is a kernel of libnbp_hip.so and nothing touches the disk inside the loop.

Per step: coverage (device counter, no sync) -> un-project the current frame -> fused 6-channel
map accumulation + trajectory channel -> replan test (mesh segment query) -> NBP forward ->
[replan: fusion, scoring, all-edges mask, host search] -> move (4 poses, ONE raster launch) ->
un-project the 4 supervision frames.  The cloud and its size live on the device.

Deliberate deviations (documented in DESIGN.md): the reference draws its 5 % sub-samples and
random headings from unseeded global generators (note R of the survey) -- here they are seeded;
obtain_depth's discarded outputs (long_term_utils.py:50-155) are not computed; the reference's
unbound-`next_idx` crash when no path exists (nbp_planning.py:255-258) becomes "turn in place".
"""
from __future__ import annotations

import json
import os
import random
import time

import numpy as np
import torch

from .. import _lib
from ..networks.nbp_model import NBP
from ..simulator import scene as sim_scene
from ..simulator.camera import Camera
from ..utility import hipops
from ..utility import utils as hu
from ..utility.long_term_utils import LatticePlanner, compute_auc

N_POSES = 101            # range(101) at nbp_planning.py:60
_STEP_MAPS = _lib.tune("NBP_STEP_MAPS", "1") == "1"      # A/B: 0 = the step's map stage as separate reference-API calls
# the eval forward as a replayed hipGraph (packing.ForwardGraph): 0 = never, 1 = a single rollout's B = 1 forward, 2 = also the
# lock-step groups' batched forwards
_FWD_GRAPH = int(_lib.tune("NBP_FWD_GRAPH", "1"))
# Rollout.step: the step's forward on a stream of its own, behind the map stage by an event.  The reference runs the network at
# every step and reads its output only when it replans (nbp_planning.py:166 / :252): a step that does not replan goes on to the move
# and the next observation while its forward (every kernel of it, on the same input) is still running.  0 = everything in stream order.
_STEP_OVERLAP = _lib.tune("NBP_STEP_OVERLAP", "1") == "1"
# the step's maps from the tile-binned shadow copy of the cloud (utils.CloudBins; bit-identical maps): 0 = the append-order kernel
_MAP_BINS = _lib.tune("NBP_MAP_BINS", "1") == "1"
# single rollout: the un-projection launches file the points they append into the bins and clear the maps, so that the step's map
# build is ONE launch (nbp_step_maps_prefiled_f32; bit-identical maps): 0 = bin_append_kernel + map_binned_kernel per build
_MAP_PREFILED = _lib.tune("NBP_MAP_PREFILED", "1") == "1"
_GC_FREEZE = _lib.tune("NBP_GC_FREEZE", "1") == "1"


def _settle_gc():
    """A rollout's setup leaves ~10^5 long-lived Python objects behind (the lattice planner's node / edge tables); CPython's
    generational collector walks all of them in every full collection -- 70-80 ms, once or twice per 100-step rollout, i.e. as long
    as 80 exploration steps (tools/diag/single_step_spikes.py: 1140 steps/s with the collector off, 450-500 with it on over steps
    40-100).  After setup they are collected once and then FROZEN (gc.freeze: moved to the permanent generation, no longer walked;
    reference counting still frees them).  NBP_GC_FREEZE=0: off."""
    if _GC_FREEZE:
        import gc
        gc.collect()
        gc.freeze()


class RolloutState:
    """Device-resident rollout buffers (cloud, counters, per-step coverage counts)."""

    def __init__(self, device, capacity=3_400_000, grid=256):
        self.device = device
        self.cloud = torch.zeros(capacity, 3, dtype=torch.float32, device=device)
        self.cloud_rgb = torch.zeros(capacity, 3, dtype=torch.float32, device=device)        # full_pc_colors (ref :39)
        self.cloud_count = torch.zeros(1, dtype=torch.int64, device=device)
        self.coverage_counts = torch.zeros(N_POSES, 2, dtype=torch.int32, device=device)
        self.maps6 = torch.zeros(6, grid, grid, dtype=torch.float32, device=device)
        self.net_in = torch.zeros(1, 5, grid, grid, dtype=torch.float32, device=device)
        self.bins = None             # utils.CloudBins of `cloud` (made by the rollout: the tile grid needs the scene's extent)
        self.frames_appended = 0     # frames un-projected into the cloud so far (host bound of its size: <= that many x per-frame keep)
        self.frames_filed = 0        # ... of which the bins have seen (filed by the appending launch, or caught up by a two-launch build)
        self._overlap = None         # forward_overlap(): two network inputs / streams / event pairs, used alternately

    def forward_overlap(self):
        """Two network-input buffers with a forward stream and an event pair each (Rollout.step alternates between them, so that
        the forward of step t may still be reading its input while step t + 1 builds the next one)."""
        if self._overlap is None:
            main = torch.cuda.current_stream(self.device)
            ins = [self.net_in, torch.zeros_like(self.net_in)]
            # streams that really run beside `main` and beside each other (HIP maps streams onto four hardware queues; a forward
            # stream that shares main's queue serialises the step: 1100 -> 890 steps/s, seen in about one bench run in eight)
            streams, self.overlap_info = _concurrent_streams(self.device, len(ins), against=(main,))
            # (a high-priority third stream for the forward of a replanning step -- the one the step waits for, ahead of the dead
            # forwards still running beside it -- was measured: 1142 / 1157 against 1135 / 1138 steps/s, within noise; not kept)
            for f in streams:
                f.wait_stream(main)
            self._overlap = {"net_in": ins, "streams": streams, "maps": [torch.cuda.Event() for _ in ins],
                             "done": [torch.cuda.Event() for _ in ins], "used": [False for _ in ins]}
        return self._overlap


def setup_test_camera(params, mesh, start_cam_idx, settings, device, seed=0):
    """macarons/testers/scene.py:410-488: build the camera, step to the first collision-free
    neighbour of the start pose, capture, then walk to the start pose capturing 4 frames."""
    cam = Camera(settings.camera.x_min, settings.camera.x_max, settings.camera.pose_l, settings.camera.pose_w,
                 settings.camera.pose_h, settings.camera.pose_n_elev, settings.camera.pose_n_azim,
                 params.n_interpolation_steps, params.zfar, params.image_height, params.image_width, device,
                 params.gathering_factor, params.sensor_range, seed=seed,
                 ambient_light_intensity=getattr(params, "ambient_light_intensity", 0.85),
                 contrast_factor=getattr(settings.camera, "contrast_factor", 1.0))
    start = tuple(int(v) for v in start_cam_idx)
    neigh = cam.get_neighboring_poses(start)
    segs = torch.from_numpy(np.stack([np.concatenate([cam.pose_from_idx(n)[:3], cam.pose_from_idx(start)[:3]])
                                      for n in neigh]).astype(np.float32)).to(device)
    hit = hipops.segments_hit_mesh(mesh.verts, mesh.faces, segs).cpu().numpy()
    free = [n for n, h in zip(neigh, hit) if not h]
    first = free[0] if free else neigh[0]          # the reference raises NameError when none is free
    cam.initialize_camera(first)
    cam.capture_image(mesh)
    cam.move_and_capture(mesh, start)
    return cam


class Rollout:
    """One exploration rollout, steppable (bench.py times K consecutive ``step()`` calls)."""

    def __init__(self, params, nbp, camera, gt_scene_pc, mesh, mesh_for_check, y_bins, device, state=None, seed=0,
                 grid=256):
        self.params, self.nbp, self.camera, self.mesh, self.mesh_for_check = params, nbp, camera, mesh, mesh_for_check
        self.y_bins, self.device = y_bins, device
        # the reference hard-codes 256 / 64 / +-40 (nbp_planning.py:43-45); other grids keep 0.3125 units per pixel
        self.S, self.V, self.grid_range = grid, grid // 4, (-40 * grid // 256, 40 * grid // 256)
        self.st = state or RolloutState(device, grid=grid)
        # a state taken over from an earlier rollout may still have that rollout's side-stream forwards in flight on its input
        # buffers: this stream waits for them before the state is overwritten
        self._wait_forwards()
        self.st.cloud_count.zero_()
        self.st.coverage_counts.zero_()
        self.st.frames_appended = 0
        self.st.frames_filed = 0
        if _MAP_BINS:
            vh = np.asarray(mesh.verts_host, np.float32)
            lo, hi = (float(vh[:, 0].min()), float(vh[:, 2].min())), (float(vh[:, 0].max()), float(vh[:, 2].max()))
            b = self.st.bins
            if b is None or (tuple(b.lo), tuple(b.hi), b.capacity) != (lo, hi, self.st.cloud.shape[0]):
                self.st.bins = hu.CloudBins(lo, hi, self.st.cloud.shape[0], device)
            else:
                b.reset()                     # the cloud starts from zero points again
        else:
            self.st.bins = None
        self.rng = random.Random(seed)
        self.planner = LatticePlanner(camera, mesh_for_check, device, self.V, self.S, self.grid_range, rng=self.rng)
        self.gt = gt_scene_pc.contiguous()
        self.bbox = (self.gt.min(0).values.tolist(), self.gt.max(0).values.tolist())
        self.cov_plan = hipops.CoveragePlan(self.gt, 1.0, 2, self.bbox)          # GT sorted once per rollout
        self.path, self.path_record = [], 0
        self.collision_list, self.passable_list, self.idx_history = [], [], []
        self.step_seed = seed * 1_000_003
        self.pose_i = 0
        self.n_replans = 0
        _settle_gc()

    # The step is split in enqueue-only halves so that MultiRollout can batch the NBP forward of several
    # rollouts and share one stream synchronisation per step; step() is the single-rollout composition.
    def pre(self, net_in=None):
        """S2-S8: coverage, un-projection of the current frame, maps, replan decision.  No host sync."""
        st, S = self.st, self.S
        net_in = st.net_in if net_in is None else net_in
        # the bins are in step with the cloud (every frame so far was filed): this step's un-projection files its points too and
        # clears the maps, and the build below is the page launch alone
        prefiled = _MAP_PREFILED and _STEP_MAPS and st.bins is not None and st.frames_filed == st.frames_appended
        self.pre_coverage()
        self.pre_unproject(clear=(st.maps6, net_in[0, 4]) if prefiled else None)
        prefiled = prefiled and st.frames_filed == st.frames_appended
        # S5-S7 in one call: six maps, trajectory channel, network input (was seven launches: accumulate_step_maps,
        # transform_points_to_n_pieces, map_points_to_n_imgs and two copies)
        if _STEP_MAPS:
            full_pc, n_upper, n_dev, pose, y_bins, traj_dev, n_old, fresh, bins = self.maps_item()
            hu.step_maps(full_pc, pose, y_bins, S, self.grid_range, traj_dev, n_old, fresh, st.maps6, net_in[0], n_dev=n_dev,
                         bins=bins, n_upper=n_upper, prefiled=prefiled)
            if bins is not None:
                st.frames_filed = st.frames_appended   # (a two-launch build files whatever the store had not seen)
            self.traj_img = net_in[0, 4]               # stays valid until this rollout's next pre()
        else:
            hu.accumulate_step_maps(st.cloud, self.pose, self.y_bins, S, self.grid_range, n_dev=st.cloud_count, out=st.maps6)
            traj2d = hu.transform_points_to_n_pieces(self.camera.trajectory_points(), self.pose)
            self.traj_img = hu.map_points_to_n_imgs(traj2d, (S, S), self.grid_range)
            net_in[0, :4] = st.maps6[:4]
            net_in[0, 4] = self.traj_img[0]
        self.pre_decide()

    def coverage_item(self):
        st, pose_i = self.st, self.pose_i
        return (self.cov_plan, st.cloud, st.coverage_counts[pose_i % N_POSES], st.cloud_count, st.cloud.shape[0],
                self.step_seed + 7 * pose_i, pose_i < N_POSES)

    def pre_observe(self):
        """S2-S4: coverage of the cloud so far, un-projection of the current frame into it."""
        self.pre_coverage()
        self.pre_unproject()

    def pre_coverage(self):
        st, pose_i = self.st, self.pose_i
        self.cov_plan.count(st.cloud, st.coverage_counts[pose_i % N_POSES], n_dev=st.cloud_count, n=st.cloud.shape[0],
                            seed=self.step_seed + 7 * pose_i, out_is_zero=pose_i < N_POSES)

    def _filing(self, depth):
        """The bins, when this un-projection call may file into them (in step with the cloud, three-launch form), else None."""
        st = self.st
        ok = _MAP_PREFILED and st.bins is not None and st.frames_filed == st.frames_appended and hipops.unproject_files(depth, None)
        return st.bins if ok else None

    def pre_unproject(self, clear=None):
        st, camera, params, pose_i = self.st, self.camera, self.params, self.pose_i
        depth, cams = camera.frames_batch([-1])
        colour = camera.colour_source([-1])
        bins = self._filing(depth)
        hipops.unproject_append(depth, None, cams, st.cloud, st.cloud_count, params.gathering_factor,
                                params.sensor_range, seed=self.step_seed + 11 * pose_i,
                                cloud_rgb=st.cloud_rgb if colour else None, bins=bins, clear=clear if bins is not None else None, **colour)
        st.frames_appended += 1
        if bins is not None:
            st.frames_filed += 1
        self.pose, _ = camera.get_pose_from_idx(camera.cam_idx)

    def maps_item(self):
        """The map stage's arguments (utils.step_maps / step_maps_batch): (cloud, host upper bound of its size, device size,
        pose, y_bins, trajectory history, n_old, fresh positions, bins).  The bound counts the frames actually un-projected into
        the cloud (a frame keeps at most int(H W gathering_factor) pixels); it only sizes grids -- the kernels read the device
        count, and the binned build walks every page whatever the bound (a guessed bound must not drop points)."""
        traj_dev, n_old, fresh = self.camera.trajectory_pending()
        per_frame = int(self.params.image_height * self.params.image_width * self.params.gathering_factor) + 1
        n_upper = self.st.cloud.shape[0] if self.st.bins is None else min(self.st.cloud.shape[0], self.st.frames_appended * per_frame)
        return self.st.cloud, n_upper, self.st.cloud_count, self.pose, self.y_bins, traj_dev, n_old, fresh, self.st.bins

    def pre_decide(self):
        """S8: does this step replan?  Host only."""
        camera, pose_i = self.camera, self.pose_i
        path = self.path
        if pose_i == 0 or not path or self.path_record + 1 > len(path):
            self.need_replan = True
        else:
            nxt = path[self.path_record]
            self.need_replan = self.planner.edge_hits_mesh(camera.cam_idx[:3], nxt[:3])
            if self.need_replan:
                cur3, nxt3 = list(camera.cam_idx[:3]), list(nxt[:3])
                self.collision_list += [[cur3, nxt3], [nxt3, cur3], list(path[-1][:3])]
        if len(self.idx_history) >= 2:
            p1, p2 = list(self.idx_history[-1][:3]), list(self.idx_history[-2][:3])
            self.passable_list += [[p1, p2], [p2, p1]]

    def plan_enqueue(self, out1, out2):
        if self.need_replan:
            self.n_replans += 1
            self.path_record = 0
            self.planner.replan_enqueue(self.pose, out1, out2, self.st.maps6, self.traj_img, self.collision_list)

    def plan_finish(self):
        if self.need_replan:
            self.path = self.planner.replan_finish(self.collision_list, self.passable_list)

    def post(self):
        """S10-S14: next pose, move (4 poses, one raster launch), un-project the supervision frames."""
        st, camera, params = self.st, self.camera, self.params
        next_idx = self.post_choose()
        camera.move_and_capture(self.mesh, next_idx)
        depth, cams = camera.frames_batch([-5, -4, -3, -2])
        colour = camera.colour_source([-5, -4, -3, -2])
        bins = self._filing(depth)
        hipops.unproject_append(depth, None, cams, st.cloud, st.cloud_count, params.gathering_factor,
                                params.sensor_range, seed=self.step_seed + 11 * self.pose_i + 5,
                                cloud_rgb=st.cloud_rgb if colour else None, bins=bins, **colour)
        st.frames_appended += 4
        if bins is not None:
            st.frames_filed += 4
        self.post_finish()

    def post_choose(self):
        """S10: the next lattice pose (host only)."""
        camera, path = self.camera, self.path
        if not path or self.path_record >= len(path):
            next_idx = list(camera.cam_idx)
            next_idx[4] = self.rng.randrange(8)
            path = []
        else:
            next_idx = list(path[self.path_record])
            if tuple(next_idx) in {tuple(h) for h in self.idx_history}:
                next_idx[4] = self.rng.randrange(8)
        self.path = path
        self.idx_history.append(tuple(camera.cam_idx))
        return next_idx

    def post_finish(self):
        self.path_record += 1
        self.pose_i += 1

    def unproject_item(self, which, seed):
        """Arguments of hipops.unproject_append_batch for frames `which` of this rollout (None: the camera renders eager colours --
        the caller falls back to the single call)."""
        camera, st = self.camera, self.st
        if camera._rgb_ring is not None:
            return None
        slots = [camera.frames[w][2] for w in which]
        hw4 = camera.image_height * camera.image_width * 4
        z0 = camera._zbuf_ring.data_ptr()
        depth = [z0 + k * hw4 for k in slots]                  # frame pointers (the ring's frames are [H,W] fp32, 16-byte aligned)
        cams = np.stack([camera.frames[w][1] for w in which]).astype(np.float32)
        shade = None
        if camera._zface_ring is not None:
            m = camera._mesh
            f0 = camera._zface_ring.data_ptr()
            shade = ([f0 + 2 * k * hw4 for k in slots], m.verts, m.faces, m.colors, camera.ambient)
        return (self, depth, cams, st.cloud, st.cloud_count, seed, st.cloud_rgb if shade else None, shade)

    def step(self):
        static = getattr(self.nbp, "forward_static", None) if _FWD_GRAPH >= 1 else None
        if static is not None and _STEP_OVERLAP and _STEP_MAPS and not self.nbp.training:
            return self._step_forward_on_its_own_stream(static)
        self.pre()
        with torch.no_grad():          # S9: one NBP forward per step (the reference also runs it without replanning, :252)
            # net_in is a persistent tensor: the ~60 launches of a B = 1 forward replay as one hipGraph (bit-identical outputs)
            out1, out2 = static(self.st.net_in) if static is not None else self.nbp(self.st.net_in)
        self.plan_enqueue(out1, out2)
        if self.need_replan:
            torch.cuda.current_stream().synchronize()
        self.plan_finish()
        self.post()

    def _step_forward_on_its_own_stream(self, static):
        """step() with the forward (the replayed hipGraph of this input buffer) on a side stream: same kernels, same inputs, same
        results; only a replanning step waits for it."""
        st = self.st
        ov = st.forward_overlap()
        slot = self.pose_i & 1
        main = torch.cuda.current_stream(self.device)
        net_in, fwd = ov["net_in"][slot], ov["streams"][slot]
        if ov["used"][slot]:
            main.wait_event(ov["done"][slot])          # the forward of two steps ago has read this buffer
        st.net_in = net_in                             # (the input of the latest step, whichever buffer it is)
        self.pre(net_in)
        ov["maps"][slot].record(main)
        with torch.cuda.stream(fwd), torch.no_grad():
            fwd.wait_event(ov["maps"][slot])
            out1, out2 = static(net_in)
            ov["done"][slot].record(fwd)
        ov["used"][slot] = True
        if self.need_replan:
            main.wait_event(ov["done"][slot])
            self.plan_enqueue(out1, out2)
            main.synchronize()
            self.plan_finish()
        self.post()

    def _wait_forwards(self):
        ov = getattr(self.st, "_overlap", None)
        if ov is not None:
            main = torch.cuda.current_stream(self.device)
            for k, (used, ev) in enumerate(zip(ov["used"], ov["done"])):
                if used:
                    main.wait_event(ev)
                    ov["used"][k] = False          # (a later step on this slot records the event again before anyone waits on it)

    def finish(self):
        """The caller's stream waits for the forwards still in flight (before the network's weights or the state are reused).
        Idempotent; coverage_evolution() -- what every driver ends a rollout with -- calls it, and a Rollout that takes over a used
        state waits in its constructor, so a loop over step() that never calls finish() is safe as well."""
        self._wait_forwards()

    def coverage_evolution(self, n):
        self.finish()
        counts = self.st.coverage_counts[:n].cpu().numpy()
        G = np.float32(len(self.gt))
        return [float(np.float32(c) / G) for c in counts[:, 0]]


def _concurrent_streams(device, n, tries=24, cycles=300_000, against=(), priority=0):
    """n torch streams whose kernels the runtime really runs side by side, and how that was established.

    HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) when they are first used, and two streams on one
    queue run their kernels strictly one after the other -- the lock-step's two groups then stop overlapping.  Measured: after
    ANY hipGraph capture in the process (the single-rollout path's ForwardGraph) the next two pool streams land on one queue and
    the 48-rollout lock-step loses 6 % (profiles/r04/stream_queue_collision.txt).  So the streams are not taken on trust: a
    candidate is kept only if a short spin kernel on it overlaps with one on every stream already chosen (and on every stream of
    `against`: the single rollout's forward streams must run beside the stream the step itself is enqueued on)."""
    def overlap(a, b):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        torch.cuda.synchronize(device)
        with torch.cuda.stream(a):
            ev[0].record()
            torch.cuda._sleep(cycles)
            ev[1].record()
        with torch.cuda.stream(b):
            torch.cuda._sleep(cycles)
            ev[2].record()
        torch.cuda.synchronize(device)
        return ev[0].elapsed_time(ev[2]) < 1.5 * ev[0].elapsed_time(ev[1])

    if not hasattr(torch.cuda, "_sleep"):          # (a torch without the spin kernel: the streams are taken as they come)
        return [torch.cuda.Stream(device, priority=priority) for _ in range(n)], {"streams_tested": 0, "concurrent": None}
    chosen, tested = [], 0
    while len(chosen) < n and tested < tries:
        st = torch.cuda.Stream(device, priority=priority)
        tested += 1
        with torch.cuda.stream(st):
            torch.cuda._sleep(1)                   # first use: the stream gets its hardware queue here
        if all(overlap(c, st) for c in list(against) + chosen):
            chosen.append(st)
    ok = len(chosen) == n
    while len(chosen) < n:                         # (never seen: fewer than n distinct queues among `tries` streams)
        chosen.append(torch.cuda.Stream(device, priority=priority))
    return chosen, {"streams_tested": tested, "concurrent": ok}


class MultiRollout:
    """R independent rollouts (different scenes / start poses) on one GPU ("B_rollout
    concurrent rollouts per rank").  They are split in two groups that are software-pipelined: while the GPU
    runs one group's batched NBP forward (B = R/2 in the GEMM M dimension) the host finishes the other
    group's replanning (event wait + search) and enqueues its move / raster / un-projection, so neither
    side idles.  Each group has its own HIP stream: the small latency-bound kernels of one group's step (coverage,
    un-projection, map accumulation, raster, planner) run underneath the other group's convolutions instead of
    in front of them.  Each rollout's results are identical to running it alone (tests/test_gpu_rollout.py)."""

    def __init__(self, rollouts, nbp, device, grid=None, streams=True, n_groups=None, elide_dead_forward=False):
        self.rollouts, self.nbp = list(rollouts), nbp
        # The reference runs the network at every step and uses its output only when it replans (nbp_planning.py:166 / :252).
        # Default (False): every rollout's map goes through the forward at every step, as there.  True (reported beside the
        # headline, never as it): only the replanning rollouts' maps are forwarded -- same trajectories, same coverage
        # (tests/test_gpu_rollout.py::test_dead_forward_elision_changes_nothing).
        self.elide_dead_forward = bool(elide_dead_forward)
        self._sub = {}
        self.device, self._packed = device, None
        grid = grid or self.rollouts[0].S
        R = len(self.rollouts)
        # two pipeline groups, and more once a group would exceed 24 rollouts (measured: 48 = 2 x 24 is the best split of one
        # GPU; groups of 30+ lose 15 % in bench.py's configuration, 3 x 24 is level with 2 x 24)
        n_groups = n_groups or int(_lib.tune("NBP_ROLLOUT_GROUPS", "0")) or max(2, (R + 23) // 24)
        n_groups = max(1, min(n_groups, R))
        per = (R + n_groups - 1) // n_groups
        self.groups = [self.rollouts[i:i + per] for i in range(0, R, per)]
        self.net_in = [torch.zeros(len(g), 5, grid, grid, dtype=torch.float32, device=device) for g in self.groups]
        # the rollouts' map stacks as slices of one tensor per group: the group's map stage is ONE batched launch
        # (NBP_STEP_BATCH=0: one launch per rollout, the A/B switch)
        self.batched = _lib.tune("NBP_STEP_BATCH", "1") == "1" and _STEP_MAPS
        # which stages of the group's step go as one launch each (A/B: NBP_STEP_BATCH_STAGES=maps,coverage,...)
        self.batch_stages = set(_lib.tune("NBP_STEP_BATCH_STAGES", "maps,coverage,unproject,raster,replan").split(","))
        self.maps6 = [torch.zeros(len(g), 6, grid, grid, dtype=torch.float32, device=device) for g in self.groups]
        self._out1_pin = [torch.empty(len(g), 8, grid // 4, grid // 4, dtype=torch.float32).pin_memory() for g in self.groups]
        self._plan_event = [None] * len(self.groups)
        for g, m6 in zip(self.groups, self.maps6):
            for i, r in enumerate(g):
                r.st.maps6 = m6[i]
        # the replanning results of a group's rollouts as rows of ONE buffer pair per group (one device -> host copy per group and
        # step); rollouts whose lattices differ in size keep their own buffers
        self._res = []
        for g in self.groups:
            nb = {r.planner.result_bytes() for r in g}
            if len(nb) == 1:
                n = nb.pop()
                devb = torch.empty(len(g), n, dtype=torch.uint8, device=device)
                pinb = torch.empty(len(g), n, dtype=torch.uint8).pin_memory()
                for i, r in enumerate(g):
                    r.planner.share_result_buffers(devb[i], pinb[i])
                self._res.append((devb, pinb))
            else:
                self._res.append(None)
        self.inflight = [False] * len(self.groups)
        # Streams.  One HIP stream per group carries the group's small kernels and its batched forward.
        # NBP_ROLLOUT_STREAMS = k > 0 (A/B switch) adds k side streams per group for the rollouts' ~20 small kernels per step
        # (rollout i on side stream i % k), chained to the forward by one event per side stream.  Measured on MI355X (16
        # rollouts, split path): 1843-1883 steps/s without side streams, 1380-1394 with k = 1, 1772-1800 with k = 2,
        # 1667-1683 with k = 4 -- every cross-stream event dependency costs more than the overlap returns (k = 1 adds
        # nothing but the two dependencies per group and lock-step and loses 2.9 ms of 8.6), and the convolutions hold
        # every CU's registers and LDS, so a small kernel that "overlaps" takes CUs away from them anyway.
        main = torch.cuda.current_stream(device)
        multi = streams and len(self.groups) >= 2
        k = int(_lib.tune("NBP_ROLLOUT_STREAMS", "0")) if multi else 0
        self.stream_check = None
        if multi:
            self.fwd_streams, self.stream_check = _concurrent_streams(device, len(self.groups))
            self.side = [[torch.cuda.Stream(device) for _ in range(min(k, len(g)))] for g in self.groups]
            for st in self.fwd_streams + [x for g in self.side for x in g]:
                st.wait_stream(main)          # the rollouts were built on the caller's stream
        else:
            self.fwd_streams = [main for _ in self.groups]
            self.side = [[] for _ in self.groups]
        self.streams = self.fwd_streams
        self.ev_pre = [[torch.cuda.Event() for _ in sd] for sd in self.side]
        self.ev_plan = [[torch.cuda.Event() for _ in (sd or g)] for sd, g in zip(self.side, self.groups)]
        self.ev_fwd = [torch.cuda.Event() for _ in self.groups]

    def _launch(self, gi):
        grp, net_in, fwd, side = self.groups[gi], self.net_in[gi], self.fwd_streams[gi], self.side[gi]
        if not side:
            # one stream per group: a single stream guard around the whole group (a guard per rollout is ~10 us of host time)
            with torch.cuda.stream(fwd):
                if self.batched:
                    self._pre_group(gi)
                else:
                    for i, r in enumerate(grp):
                        r.pre(net_in[i:i + 1])
                rows = None
                if self.elide_dead_forward and self.batched and "replan" in self.batch_stages and self._packed is not None:
                    out1, out2, rows = self._forward_replanning_only(gi)
                else:
                    with torch.no_grad():
                        out1, out2 = self._forward(net_in)
                if self.batched and "replan" in self.batch_stages:
                    self._plan_group(gi, out1, out2, rows)
                else:
                    for i, r in enumerate(grp):
                        r.plan_enqueue(out1[i], out2[i])
                        self.ev_plan[gi][i].record()
            self.inflight[gi] = True
            return
        k = len(side)
        for si, st in enumerate(side):
            with torch.cuda.stream(st):
                for i in range(si, len(grp), k):
                    grp[i].pre(net_in[i:i + 1])
                self.ev_pre[gi][si].record()
        with torch.cuda.stream(fwd):
            for ev in self.ev_pre[gi]:
                fwd.wait_event(ev)
            with torch.no_grad():
                out1, out2 = self._forward(net_in)
            self.ev_fwd[gi].record()
        for si, st in enumerate(side):
            with torch.cuda.stream(st):
                st.wait_event(self.ev_fwd[gi])     # also orders the next pre() after this forward's read of net_in
                out1.record_stream(st); out2.record_stream(st)
                for i in range(si, len(grp), k):
                    grp[i].plan_enqueue(out1[i], out2[i])
                self.ev_plan[gi][si].record()
        self.inflight[gi] = True

    def _pre_group(self, gi):
        """Rollout.pre for every rollout of the group, each latency-bound stage as ONE batched launch (identical results)."""
        grp, net_in = self.groups[gi], self.net_in[gi]
        p0, stages = grp[0].params, self.batch_stages
        if "coverage" in stages:
            hipops.coverage_count_batch([r.coverage_item() for r in grp])
        else:
            for r in grp:
                r.pre_coverage()
        items = [r.unproject_item([-1], r.step_seed + 11 * r.pose_i) for r in grp] if "unproject" in stages else [None]
        if all(it is not None for it in items):
            hipops.unproject_append_batch(items, p0.image_height, p0.image_width, 1, p0.gathering_factor, p0.sensor_range)
            for r in grp:
                r.st.frames_appended += 1
                r.pose, _ = r.camera.get_pose_from_idx(r.camera.cam_idx)
        else:
            for r in grp:
                r.pre_unproject()
        if "maps" in stages:
            hu.step_maps_batch([r.maps_item() for r in grp], grp[0].S, grp[0].grid_range, self.maps6[gi], net_in)
        for i, r in enumerate(grp):
            if "maps" not in stages:
                full_pc, n_upper, n_dev, pose, y_bins, traj_dev, n_old, fresh, bins = r.maps_item()
                hu.step_maps(full_pc, pose, y_bins, r.S, r.grid_range, traj_dev, n_old, fresh, r.st.maps6, net_in[i], n_dev=n_dev,
                             bins=bins, n_upper=n_upper)
            r.traj_img = net_in[i, 4]
            r.pre_decide()

    def _forward_replanning_only(self, gi):
        """elide_dead_forward: the forward over the maps of the rollouts that replan this step only -> (out1, out2, {rollout
        index in the group: row}); (None, None, {}) when none does.  The rows are gathered on the device (index list through a
        pinned buffer: no pageable copy in the loop); the workspace is the full group's."""
        from ..networks import packing
        grp, net_in = self.groups[gi], self.net_in[gi]
        need = [i for i, r in enumerate(grp) if r.need_replan]
        if not need:
            return None, None, {}
        if len(need) == len(grp):
            with torch.no_grad():
                out1, out2 = self._forward(net_in)
            return out1, out2, None
        st = self._sub.get(gi)
        if st is None:
            n, S = net_in.shape[0], net_in.shape[-1]
            prec = self._packed.precision
            nbytes = max(int(getattr(_lib.lib(), packing._FWD[prec][1])(b, S)) for b in range(1, n + 1))
            st = self._sub[gi] = {"pin": [torch.zeros(n, dtype=torch.int64).pin_memory() for _ in range(4)], "k": 0,
                                  "idx": torch.zeros(n, dtype=torch.int64, device=self.device), "x": torch.empty_like(net_in),
                                  "ws": torch.empty(nbytes, dtype=torch.uint8, device=self.device)}
        k = len(need)
        pin = st["pin"][st["k"] % 4]
        st["k"] += 1
        pin[:k] = torch.tensor(need, dtype=torch.int64)
        st["idx"][:k].copy_(pin[:k], non_blocking=True)
        x = st["x"][:k]
        torch.index_select(net_in, 0, st["idx"][:k], out=x)
        with torch.no_grad():
            out1, out2 = packing.forward_packed(self._packed, x, ws=st["ws"])
        return out1, out2, {i: row for row, i in enumerate(need)}

    def _plan_group(self, gi, out1, out2, rows=None):
        """Rollout.plan_enqueue for the group: the replanning rollouts' GPU half in two launches, their results in one copy each
        plus ONE copy of the group's value maps; one event for all of them.  rows: rollout index -> row of out1 / out2 (None:
        the identity)."""
        grp = self.groups[gi]
        need = [(i, r) for i, r in enumerate(grp) if r.need_replan]
        if need:
            r0 = grp[0]
            pin = self._out1_pin[gi]
            items = []
            row = (lambda i: i) if rows is None else rows.__getitem__
            for i, r in need:
                r.n_replans += 1
                r.path_record = 0
                items.append(r.planner.replan_item(r.pose, out1[row(i)].reshape(8, r.V, r.V), out2[row(i)].reshape(r.S, r.S), r.st.maps6,
                                                   r.traj_img.reshape(r.S, r.S), r.collision_list))
            hipops.replan_batch(items, r0.S, r0.V, r0.grid_range)
            pin[:out1.shape[0]].copy_(out1, non_blocking=True)
            if self._res[gi] is not None:
                self._res[gi][1].copy_(self._res[gi][0], non_blocking=True)        # every rollout's (scores, valid, blocked) rows at once
            for i, r in need:
                r.planner.replan_copy_back(r.pose, pin[row(i)].reshape(8, r.V, r.V))
        ev = self.ev_plan[gi][0]
        ev.record()
        self._plan_event[gi] = ev

    def _post_group(self, grp):
        """Rollout.post for the rollouts of a group: their moves are rendered in ONE batched rasteriser call and their
        supervision frames un-projected in one (identical results)."""
        p0 = grp[0].params
        H, W = p0.image_height, p0.image_width
        stages = self.batch_stages
        can_raster = "raster" in stages and all(r.camera.deferred_colours(r.mesh) for r in grp)
        pend = []
        for r in grp:
            cams = r.camera.move_poses(r.post_choose())
            if can_raster:
                out, zf, slot = r.camera.capture_begin(r.mesh, cams)
                pend.append((r, cams, out, zf, slot))
            else:
                r.camera.capture_images(r.mesh, cams)
        if can_raster:
            hipops.raster_zface_batch([(r, r.mesh.verts, r.mesh.faces, cams, out, zf) for r, cams, out, zf, _ in pend], H, W,
                                      len(pend[0][1]))
            for r, cams, out, _, slot in pend:
                r.camera.capture_commit(out, cams, slot)
        which = [-5, -4, -3, -2]
        items = [r.unproject_item(which, r.step_seed + 11 * r.pose_i + 5) for r in grp] if "unproject" in stages else [None]
        if all(it is not None for it in items):
            hipops.unproject_append_batch(items, H, W, 4, p0.gathering_factor, p0.sensor_range)
            for r in grp:
                r.st.frames_appended += 4
        else:
            for r in grp:
                st, camera = r.st, r.camera
                depth, cams = camera.frames_batch(which)
                colour = camera.colour_source(which)
                hipops.unproject_append(depth, None, cams, st.cloud, st.cloud_count, p0.gathering_factor, p0.sensor_range,
                                        seed=r.step_seed + 11 * r.pose_i + 5, cloud_rgb=st.cloud_rgb if colour else None, **colour)
                st.frames_appended += 4
        for r in grp:
            r.post_finish()

    def _forward(self, net_in):
        """The rollouts evaluate a frozen network: its packed weights are looked up (and their staleness checked: 327 tensor
        versions) once per lock-step, not once per forward."""
        if self._packed is None:
            return self.nbp(net_in)
        if _FWD_GRAPH >= 2:
            return self.nbp.forward_static(net_in)
        from ..networks import packing
        return packing.forward_packed(self._packed, net_in)

    def _complete(self, gi):
        grp, side = self.groups[gi], self.side[gi]
        if not side:
            if self._plan_event[gi] is not None:
                if any(r.need_replan for r in grp):
                    self._plan_event[gi].synchronize()     # one event for the group's batched replan results
                self._plan_event[gi] = None
            else:
                for i, r in enumerate(grp):
                    if r.need_replan:
                        self.ev_plan[gi][i].synchronize()      # the GPU keeps running whatever was queued after the event
            with torch.cuda.stream(self.fwd_streams[gi]):
                for r in grp:
                    r.plan_finish()
                if self.batched:
                    self._post_group(grp)
                else:
                    for r in grp:
                        r.post()
            self.inflight[gi] = False
            return
        k = len(side)
        for si, st in enumerate(side):
            mine = grp[si::k]
            if any(r.need_replan for r in mine):
                self.ev_plan[gi][si].synchronize()         # the GPU keeps running whatever was queued after the event
            with torch.cuda.stream(st):
                for r in mine:
                    r.plan_finish()
                    r.post()
        self.inflight[gi] = False

    def step(self):
        """One exploration step of every rollout (completions are deferred: a group is finished right after the
        next group has been launched, so the GPU always has another group's forward queued)."""
        G = len(self.groups)
        ensure = getattr(self.nbp, "_ensure_packed", None)
        self._packed = ensure(self.device) if ensure is not None and not self.nbp.training else None
        for gi in range(G):
            if self.inflight[gi]:
                self._complete(gi)
            self._launch(gi)
            prev = (gi - 1) % G
            if G >= 2 and self.inflight[prev] and prev != gi:
                self._complete(prev)
        if G == 1:
            self._complete(0)

    def flush(self):
        for gi in range(len(self.groups)):
            if self.inflight[gi]:
                self._complete(gi)
        main = torch.cuda.current_stream()
        for st in set(self.fwd_streams) | {x for g in self.side for x in g}:
            if st is not main:
                main.wait_stream(st)           # later work on the caller's stream sees every rollout's results


def compute_nbp_trajectory(params, nbp, camera, gt_scene_pc, mesh, mesh_for_check, n_pieces, y_bins, device,
                           test_resolution=0.05, use_perfect_depth_map=True, n_poses=N_POSES, state=None, seed=0):
    """Same name / leading arguments / return tuple as the reference (nbp_planning.py:23-361)."""
    t1 = time.time()
    nbp.eval()
    ro = Rollout(params, nbp, camera, gt_scene_pc, mesh, mesh_for_check, y_bins, device, state, seed)
    for _ in range(n_poses):
        ro.step()
    ro.finish()
    coverage_evolution = ro.coverage_evolution(n_poses)
    n_cloud = int(ro.st.cloud_count.item())
    print("Time: ", time.time() - t1)
    colors = ro.st.cloud_rgb[:n_cloud] if camera.renders_colours else None
    return coverage_evolution, camera.X_cam_history, camera.V_cam_history, ro.st.cloud[:n_cloud], colors


def load_params(path):
    """macarons/utility/utils.py:44-83: JSON -> attribute object, `_section` keys flattened away."""
    with open(path) as fh:
        raw = json.load(fh)

    def flat(d, out):
        for k, v in d.items():
            if isinstance(v, dict) and k.startswith("_"):
                flat(v, out)
            else:
                out[k] = v
        return out

    class Params:
        pass

    p = Params()
    for k, v in flat(raw, {}).items():
        setattr(p, k, v)
    return p


def list_runs(dataset, params):
    """Flattened (scene x start pose) list: the unit of scene-parallel sharding."""
    runs = []
    for si in range(len(dataset)):
        sd = dataset[si]
        st = sim_scene.Settings(sd["settings"], params.scene_scale_factor)
        for k in range(len(st.camera.start_positions)):
            runs.append((si, k))
    return runs


def build_rollout(params, nbp, dataset, run, device, test_resolution=0.05, state=None, seed=0, grid=256):
    """Scene + GT surface + camera + Rollout for one (scene, start pose) run (nbp_planning.py:414-492)."""
    si, k = run
    sd = dataset[si]
    settings = sim_scene.Settings(sd["settings"], params.scene_scale_factor)
    mesh = sim_scene.load_scene(os.path.join(dataset.data_path, sd["scene_name"], sd["obj_name"]),
                                params.scene_scale_factor, device)
    y_bins = sim_scene.y_bins_for(mesh.verts_host, 4)
    _, gt_dev = sim_scene.setup_gt_scene(params, settings, mesh, device, test_resolution, seed=seed)
    camera = setup_test_camera(params, mesh, settings.camera.start_positions[k], settings, device, seed=seed)
    ro = Rollout(params, nbp, camera, gt_dev, mesh, mesh, y_bins, device, state, seed, grid)
    ro.scene_name, ro.start = sd["scene_name"], k
    return ro


def _result(ro, n_poses):
    return {"scene": ro.scene_name, "start": ro.start, "coverage": ro.coverage_evolution(n_poses),
            "X_cam_history": ro.camera.X_cam_history.tolist(), "V_cam_history": ro.camera.V_cam_history.tolist(),
            "n_points": int(ro.st.cloud_count.item())}


def run_one(params, nbp, dataset, run, device, test_resolution=0.05, state=None, n_poses=N_POSES, seed=0):
    ro = build_rollout(params, nbp, dataset, run, device, test_resolution, state, seed)
    nbp.eval()
    for _ in range(n_poses):
        ro.step()
    ro.finish()
    return _result(ro, n_poses)


def run_many(params, nbp, dataset, runs, device, seeds, test_resolution=0.05, n_poses=N_POSES, rollouts_per_gpu=48,
             grid=256, timing=None):
    """Runs `runs` in lock-step groups of `rollouts_per_gpu` (MultiRollout); same results as run_one each.  `timing` (a dict)
    receives the wall-clock seconds of scene / GT-surface / camera setup and of the stepping (device-synchronised)."""
    import time
    out = []
    nbp.eval()
    for g0 in range(0, len(runs), rollouts_per_gpu):
        t0 = time.perf_counter()
        chunk = list(zip(runs[g0:g0 + rollouts_per_gpu], seeds[g0:g0 + rollouts_per_gpu]))
        ros = [build_rollout(params, nbp, dataset, run, device, test_resolution, None, seed, grid)
               for run, seed in chunk]
        multi = MultiRollout(ros, nbp, device)
        if timing is not None:
            torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(n_poses):
            multi.step()
        multi.flush()
        out += [_result(ro, n_poses) for ro in ros]
        if timing is not None:
            torch.cuda.synchronize(device)
            timing["build_s"] = timing.get("build_s", 0.0) + (t1 - t0)
            timing["step_s"] = timing.get("step_s", 0.0) + (time.perf_counter() - t1)
        del ros, multi
        torch.cuda.empty_cache()
    return out


def test_nbp_planning(params_file, model_file, results_json_file, numGPU, test_scenes, test_resolution=0.05,
                      use_perfect_depth_map=False, compute_collision=False, load_json=False, dataset_path=None,
                      nbp_weights=None, configs_dir=None, results_dir=None, n_poses=N_POSES, seed=8, torch_seed=9,
                      rollouts_per_gpu=48, grid_size=256, nbp_precision=None):
    """Same arguments as the reference (nbp_planning.py:364-374); `grid_size` / `nbp_precision` select
    BASELINE.json configs[4] (512 grid at the same 0.3125 units per pixel, bf16 convolutions).  Under torchrun the flattened
    (scene, start pose) runs are sharded round-robin over the ranks and the coverage curves are
    gathered with ONE all_gather over RCCL (backend "nccl" on ROCm; "gloo" on CPU-only hosts)."""
    import time
    from ..parallel_rollout import gather_results, init_distributed, shard
    t_start = time.perf_counter()
    here = os.path.dirname(os.path.abspath(__file__))
    configs_dir = configs_dir or os.path.join(here, "../../configs/macarons")
    results_dir = results_dir or os.path.join(here, "../../data")
    params = load_params(os.path.join(configs_dir, params_file))
    params.test_scenes = test_scenes
    rank, world, local_rank = init_distributed()
    device = torch.device("cuda", local_rank if world > 1 else numGPU)
    torch.cuda.set_device(device)
    random.seed(seed); np.random.seed(seed); torch.manual_seed(torch_seed)
    nbp = NBP()
    if nbp_weights and os.path.exists(nbp_weights):
        ck = torch.load(nbp_weights, map_location="cpu")
        nbp.load_state_dict(ck["model_state_dict"])
    else:
        from ..utility.synthetic import make_explorer_state_dict
        print("[nbp] no checkpoint at", nbp_weights, "-> seeded synthetic weights")
        nbp.load_state_dict(make_explorer_state_dict(torch_seed))
    nbp.to(device).eval()
    assert nbp_precision in (None, "fp32", "fp32_split", "bf16"), nbp_precision
    if nbp_precision is not None:           # None: the model's default ("fp32_split")
        nbp.conv_precision = nbp_precision
    dataset = sim_scene.SceneDataset(dataset_path, test_scenes)
    runs = list_runs(dataset, params)
    mine = shard(runs, rank, world)
    timing = {"load_s": time.perf_counter() - t_start}          # parameters, weights (packed on first use), dataset listing
    with torch.no_grad():
        results = run_many(params, nbp, dataset, mine, device, [seed + 1000 * r[0] + r[1] for r in mine],
                           test_resolution, n_poses, rollouts_per_gpu, grid_size, timing=timing)
    for run, res in zip(mine, results):
        res["run_id"] = runs.index(run)
    t_gather = time.perf_counter()
    gathered = gather_results(results, runs, rank, world, device, n_poses)
    if rank == 0:
        out = {}
        for r in gathered:
            si, k = runs[r["run_id"]]
            rec = {"coverage": r["coverage"], "auc": r["auc"]}
            for key in ("X_cam_history", "V_cam_history"):      # histories of other ranks stay in their part files
                if key in r:
                    rec[key] = r[key]
            out.setdefault(dataset[si]["scene_name"], {})[str(k)] = rec
        os.makedirs(results_dir, exist_ok=True)
        with open(os.path.join(results_dir, results_json_file), "w") as fh:
            json.dump(out, fh)
        print("Saved data about test losses in", results_json_file)
        print("All trajectories computed.")
        # one machine-readable line for bench.py's full_rollout stage: where the wall clock of a whole test run goes
        timing.update(gather_write_s=time.perf_counter() - t_gather, total_s=time.perf_counter() - t_start, runs=len(runs),
                      runs_this_rank=len(mine), n_poses=n_poses, world=world, scenes=len(dataset))
        print("[nbp] timing " + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in timing.items()}), flush=True)
    return gathered
