#!/bin/bash
# round 4: fused-launch phase clocks, and the bench by fused grid x render lanes x depth
export TMPDIR=/tmp
OUT=gpurun_out/r4s1
mkdir -p $OUT
python tools/fused_clocks.py 2>/dev/null | tee $OUT/fused_clocks.txt
PN_FUSED_GRID=128 python tools/fused_clocks.py 2>/dev/null | tee $OUT/fused_clocks_g128.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['value_unprimed'], d['config']['frames_continued_past_captured_trips'], d['breakdown_ms']['render_frame_eager'])"; }
for G in 64 96 128 160 192 256; do
  for L in 2 3; do
    echo "grid=$G lanes=$L $(PN_FUSED_GRID=$G $B --lanes $L 2>/dev/null | val)" | tee -a $OUT/sweep.txt
  done
done
for G in 64 128; do echo "grid=$G lanes=4 $(PN_FUSED_GRID=$G $B --lanes 4 2>/dev/null | val)" | tee -a $OUT/sweep.txt; done
echo "grid=128 lanes=3 depth=3 $(PN_FUSED_GRID=128 $B --lanes 3 --depth 3 2>/dev/null | val)" | tee -a $OUT/sweep.txt
echo "grid=128 lanes=3 depth=1 $(PN_FUSED_GRID=128 $B --lanes 3 --depth 1 2>/dev/null | val)" | tee -a $OUT/sweep.txt
