timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_static.py -m gpu -q -x 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-extras --config trex 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('trex', d['value'], d['ms_per_step'], d['roofline']['ms_per_frame'], d['breakdown_ms'])"
