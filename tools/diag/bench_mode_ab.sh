# the default bench (with its single-rollout stage -- a hipGraph capture -- and early window in front of the timed region) against
# the profiling form (--no-extra-stages) and against NBP_FWD_GRAPH=0, same box: does anything that runs BEFORE the timed region
# change it?  (round 4: yes -- after a graph capture two pool streams shared a hardware queue; MultiRollout now checks its streams)
set -u
OUT=gpurun_out/bench_mode_ab; mkdir -p $OUT
F="--steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-strong"
val() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[2], 'steps/s', d['value'] or d.get('value_with_numerics_knobs'), 'ms/lockstep', d['ms_per_step'], 'fwd ms', d['stages']['nbp_forward']['ms'], 'single', d['stages'].get('single_rollout_steps_per_s'), 'streams', d['stages'].get('group_streams'))" $1 "$2"; }
timeout 400 python bench.py $F > $OUT/full_a.json 2> $OUT/full_a.err; val $OUT/full_a.json "full"
timeout 400 python bench.py $F --no-extra-stages > $OUT/noextra.json 2> $OUT/noextra.err; val $OUT/noextra.json "no-extra-stages"
timeout 400 env NBP_TUNING=1 NBP_FWD_GRAPH=0 python bench.py $F > $OUT/graph0.json 2> $OUT/graph0.err; val $OUT/graph0.json "full, NBP_FWD_GRAPH=0"
timeout 400 python bench.py $F > $OUT/full_b.json 2> $OUT/full_b.err; val $OUT/full_b.json "full"
