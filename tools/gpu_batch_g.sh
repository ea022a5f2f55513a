#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02g
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_edges.py -m gpu -q -k adversarial 2>&1 | tail -3
for v in base w3 w5 w6 cf8; do
  if [ $v = base ]; then unset PN_LIB_PATH; else export PN_LIB_PATH=$PWD/pienerf_amd/lib/variants/$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 150 > $O/bench_$v.json 2> $O/e_$v
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_$v.json'))
    print('$v', d['value'], d['ms_per_step'], 'march', d['roofline']['ms_per_frame'], d['breakdown_ms']['march_per_trip'], 'eager', d['breakdown_ms']['render_frame_eager'])
except Exception as e:
    print('$v failed', e, open('$O/e_$v').read()[-300:])
PY
done
