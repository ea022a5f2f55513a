#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03g; mkdir -p $O
for cfg in "3 1" "3 2" "3 3" "2 2" "2 3"; do
  set -- $cfg
  python bench.py --lanes $1 --depth $2 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/b_$1_$2.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_$1_$2.json')); print('lanes $1 depth $2:', d['value'], d['value_unprimed'])"
done
for t in 6 7 8; do
  python bench.py --trips $t --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/b_t$t.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_t$t.json')); print('trips $t:', d['value'], d['config']['frames_continued_past_captured_trips'])"
done
