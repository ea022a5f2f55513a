#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4ab3; mkdir -p $OUT
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'])"; }
for i in 1 2; do
echo "fold refill $($B 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "nofold refill $(PN_FUSED_FOLD=0 $B 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "fold norefill $(PN_LPR_REFILL=0 $B 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "nofold norefill $(PN_FUSED_FOLD=0 PN_LPR_REFILL=0 $B 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
rocm-smi --showclocks 2>/dev/null | head -20 | tee $OUT/clocks.txt
