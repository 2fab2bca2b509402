# steps/s of the default lock-step with one switch flipped, same box, same process settings
# usage: bash tools/diag/step_ab.sh "NBP_MAP_BINS=0" "NBP_FWD_GRAPH=2" ...   (each argument = one variant; the default runs first and last)
set -u
OUT=gpurun_out/step_ab; mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-extra-stages --no-strong"
val() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[2], 'steps/s', d['value'] or d.get('value_with_numerics_knobs'), 'ms/lockstep', d['ms_per_step'], 'fwd ms', d['stages']['nbp_forward']['ms'], 'power', (d['power']['timed_region'] or {}).get('board_power_w_mean'))" $1 "$2"; }
timeout 400 $B > $OUT/default_a.json 2> $OUT/default_a.err; val $OUT/default_a.json default
i=0
for v in "$@"; do
  i=$((i+1))
  timeout 400 env NBP_TUNING=1 $v $B > $OUT/v$i.json 2> $OUT/v$i.err; val $OUT/v$i.json "$v"
done
timeout 400 $B > $OUT/default_b.json 2> $OUT/default_b.err; val $OUT/default_b.json default
