#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4grids; mkdir -p $OUT
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'])"; }
for G in 512 1024 2048 4096; do echo "march_grid=$G $(PN_MARCH_GRID=$G $B 2>/dev/null | val)" | tee -a $OUT/sweep.txt; done
for G in 256 512; do echo "tail_grid=$G $(PN_TAIL_GRID=$G $B 2>/dev/null | val)" | tee -a $OUT/sweep.txt; done
echo "depth=3 $($B --depth 3 2>/dev/null | val)" | tee -a $OUT/sweep.txt
echo "depth=1 $($B --depth 1 2>/dev/null | val)" | tee -a $OUT/sweep.txt
