#!/bin/bash
# round 4: the fused later trips against the per-trip launches, by render lanes and by the fused launch's grid
export TMPDIR=/tmp
OUT=gpurun_out/r4f2
mkdir -p $OUT
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['value_unprimed'], d['config']['frames_continued_past_captured_trips'], d['breakdown_ms']['render_frame_eager'])"; }
for L in 1 2 3; do
  echo "fused lanes=$L $($B --lanes $L 2>$OUT/err_f$L.txt | val)" | tee -a $OUT/lanes.txt
  echo "classic lanes=$L $(PN_FUSED=0 $B --lanes $L 2>$OUT/err_c$L.txt | val)" | tee -a $OUT/lanes.txt
done
for G in 128 192; do
  echo "fused grid=$G lanes=3 $(PN_FUSED_GRID=$G $B --lanes 3 2>/dev/null | val)" | tee -a $OUT/lanes.txt
done
for G in 128; do
  echo "fused grid=$G lanes=2 $(PN_FUSED_GRID=$G $B --lanes 2 2>/dev/null | val)" | tee -a $OUT/lanes.txt
done
echo "fused lanes=4 $($B --lanes 4 2>/dev/null | val)" | tee -a $OUT/lanes.txt
tail -3 $OUT/err_f3.txt
