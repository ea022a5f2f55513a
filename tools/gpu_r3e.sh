#!/bin/bash
# pipeline shape sweep: lanes x copy stream
R=$PWD; O=$R/gpurun_out/r03e; mkdir -p $O
for cfg in "3 lane" "2 copy" "3 copy" "2 lane" "3 sim" "4 lane"; do
  set -- $cfg
  python bench.py --lanes $1 --copy-on $2 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/b_$1_$2.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_$1_$2.json')); print('lanes $1 copy-on $2:', d['value'], 'steps/s', d['ms_per_step'])"
done
python bench.py --no-d2h --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/b_nod2h.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/b_nod2h.json')); print('no d2h:', d['value'])"
