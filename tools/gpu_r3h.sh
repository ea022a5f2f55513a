#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03h; mkdir -p $O
for v in 0 8192 1250; do
  PN_MARCH_GRID_LATER=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/b_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_$v.json')); print('later grid $v:', d['value'], d['value_unprimed'], d['breakdown_ms']['march_per_trip'], d['breakdown_ms']['render_frame_eager'])"
done
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "frame or march or pipelined" 2>&1 | tail -2
