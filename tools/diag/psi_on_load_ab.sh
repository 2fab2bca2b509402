# Diagnostic (GPU box): x * psi formed in the consumer's halo staging (NBP_SPLIT_PSI_ON_LOAD: 1 = levels with >= 128 channels, the default; 2 = every fused
# gate) against the gated tensor written by the gate launch and read back (0): the B = 24 forward and the lock-step, same box, alternating
set -u
for rep in 1 2; do
  for v in 1 0 2; do
    echo "== NBP_SPLIT_PSI_ON_LOAD=$v"
    NBP_TUNING=1 NBP_SPLIT_PSI_ON_LOAD=$v python tools/bench_forward.py --split --batch 24 --size 256 --reps 30 2>/dev/null | grep -E "Att|Up_conv._.*conv.0|fp32_split" | cut -c1-150
  done
done
for rep in 1 2 3; do
  for v in 1 0; do
    NBP_TUNING=1 NBP_SPLIT_PSI_ON_LOAD=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-extra-stages --no-strong 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('PSI_ON_LOAD=$v steps/s', d['value'], 'ms/lockstep', d['ms_per_step'], 'fwd ms', d['stages']['nbp_forward']['ms'])"
  done
done
