# steps/s against the number of concurrent rollouts per GPU (and pipeline groups), same box
set -u
OUT=gpurun_out/rollouts_sweep; mkdir -p $OUT
for cfg in "48 0" "40 0" "56 0" "64 0" "72 3" "48 3" "48 0"; do
  set -- $cfg
  timeout 500 env NBP_TUNING=1 NBP_ROLLOUT_GROUPS=$2 python bench.py --steps 20 --warmup 5 --rollouts-per-gpu $1 --no-cpu-baseline --no-live-traffic --no-extra-stages --no-strong > $OUT/r$1_g$2.json 2> $OUT/r$1_g$2.err
  python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print('rollouts', sys.argv[2], 'groups', sys.argv[3], 'steps/s', d['value'], 'ms/lockstep', d['ms_per_step'], 'fwd', d['stages']['nbp_forward']['batch'], d['stages']['nbp_forward']['ms'])" $OUT/r$1_g$2.json $1 $2
done
