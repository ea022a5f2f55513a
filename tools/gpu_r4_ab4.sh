#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4ab4; mkdir -p $OUT
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'])"; }
V=$PWD/pienerf_amd/lib/variants/prev_fold.so
for i in 1 2; do
echo "now fold $($B 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "prev fold $(PN_LIB_PATH=$V $B 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "now nofold $(PN_FUSED_FOLD=0 $B 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "prev nofold $(PN_LIB_PATH=$V PN_FUSED_FOLD=0 $B 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
