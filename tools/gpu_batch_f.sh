#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_shim.py tests/test_frames_gpu.py -m gpu -q -x 2>&1 | tail -15
python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/e1; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02f/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['ms_per_frame'], d['breakdown_ms'])
PY
