#!/bin/bash
# round 4: the bench lines through the reworked bench.py
export TMPDIR=/tmp
OUT=gpurun_out/r4bench
mkdir -p $OUT
( time python bench.py > $OUT/bench_chair.json 2> $OUT/bench_chair.err ) 2>&1 | grep real
tail -5 $OUT/bench_chair.err
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_chair_20steps.json 2> $OUT/bench_20.err; tail -3 $OUT/bench_20.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4bench/bench_chair.json'))
for k in ('value','value_unprimed','ms_per_step','verified','verified_note','latency_ms_per_step','two_lanes','device_resident','predicted_scaling','other_configs','breakdown_ms'):
    print(k, json.dumps(d.get(k))[:1500])
r=dict(d['roofline']); print(json.dumps({k:v for k,v in r.items() if k not in ('note','kernel')})[:3000])
n=d['network']; print(json.dumps({k:n[k] for k in ('achieved','frac','ms_per_frame','per_trip_launch_ms','all_samples_one_launch')}))
d=json.load(open('gpurun_out/r4bench/bench_chair_20steps.json')); print('20 steps', d['value'], d['value_unprimed'], d['verified'], d['roofline']['frac'])
PY
