#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03f; mkdir -p $O
for cfg in "3 host" "3 lane" "4 host" "2 host"; do
  set -- $cfg
  python bench.py --lanes $1 --copy-on $2 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/b_$1_$2.json 2> $O/b_$1_$2.err || tail -5 $O/b_$1_$2.err
  python -c "
import json; d=json.load(open('$O/b_$1_$2.json')); print('lanes $1 copy-on $2:', d['value'], 'steps/s', d['ms_per_step'])"
done
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipelined" 2>&1 | tail -2
