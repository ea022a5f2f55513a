#!/bin/bash
# round 4: one-lane rounds of the first trip's pass 1 before a ray goes to the wave-per-ray tail pass, with the later trips fused
export TMPDIR=/tmp
OUT=gpurun_out/r4lpr
mkdir -p $OUT
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], b['render_frame_eager'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'])"; }
for R in 8 16 24 32 48 64 96; do echo "lpr=$R $(PN_MARCH_LPR=$R $B 2>/dev/null | val)" | tee -a $OUT/sweep.txt; done
for G in 512 2048; do echo "lpr=24 tail_grid=$G $(PN_MARCH_LPR=24 PN_TAIL_GRID=$G $B 2>/dev/null | val)" | tee -a $OUT/sweep.txt; done
echo "latency form (lanes 3) $(PN_MARCH_LPR=0 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | val)" | tee -a $OUT/sweep.txt
