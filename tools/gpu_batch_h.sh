#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02h
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -5
python bench.py --no-cpu-baseline --no-extras --steps 150 > $O/bench_finv.json 2> $O/e_finv
python - <<PY
import json
try:
    d=json.load(open('$O/bench_finv.json'))
    print('finv', d['value'], d['ms_per_step'], 'march', d['roofline']['ms_per_frame'], d['breakdown_ms']['march_per_trip'], 'eager', d['breakdown_ms']['render_frame_eager'])
except Exception as e:
    print('failed', e, open('$O/e_finv').read()[-300:])
PY
