timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py tests/test_frames_gpu.py -m gpu -q -x 2>&1 | tail -2
for v in base split512; do
  if [ $v = base ]; then unset PN_LIB_PATH; else export PN_LIB_PATH=$PWD/pienerf_amd/lib/variants/$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['ms_per_frame'], d['breakdown_ms']['march_per_trip'], d['config']['launch'][:30])"
done
