#!/bin/bash
# round 4: the whole frame in the fused launch — tests, phase clocks, bench by grid / lanes / first-trip rounds
export TMPDIR=/tmp
OUT=gpurun_out/r4w1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -15 | tee $OUT/pytest_fused.txt
python tools/fused_clocks.py 2>/dev/null | tee $OUT/fused_clocks.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['value_unprimed'], d['config']['frames_continued_past_captured_trips'], d['breakdown_ms']['render_frame_eager'])"; }
echo "whole default lanes=3 $($B 2>$OUT/err_default.txt | val)" | tee -a $OUT/sweep.txt
tail -3 $OUT/err_default.txt
for G in 128 256; do
  for L in 2 3; do
    echo "whole grid=$G lanes=$L $(PN_FUSED_GRID=$G $B --lanes $L 2>/dev/null | val)" | tee -a $OUT/sweep.txt
  done
done
for R in 8 12 48; do echo "whole grid=128 lanes=3 arounds=$R $(PN_FUSED_AROUNDS=$R PN_FUSED_GRID=128 $B 2>/dev/null | val)" | tee -a $OUT/sweep.txt; done
echo "no-whole grid=128 lanes=3 $(PN_FUSED_WHOLE=0 PN_FUSED_GRID=128 $B 2>/dev/null | val)" | tee -a $OUT/sweep.txt
echo "whole grid=128 lanes=1 $(PN_FUSED_GRID=128 $B --lanes 1 2>/dev/null | val)" | tee -a $OUT/sweep.txt
echo "whole grid=256 lanes=1 $($B --lanes 1 2>/dev/null | val)" | tee -a $OUT/sweep.txt
