#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4ab5; mkdir -p $OUT
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'])"; }
for i in 1 2 3; do
echo "K20 fold $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "K20 nofold $(PN_FUSED_FOLD=0 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
for i in 1 2; do
echo "K200 fold $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "K200 nofold $(PN_FUSED_FOLD=0 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "lanes1 fold $(python bench.py --no-extras --no-cpu-baseline --lanes 1 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "lanes1 nofold $(PN_FUSED_FOLD=0 python bench.py --no-extras --no-cpu-baseline --lanes 1 2>/dev/null | val)" | tee -a $OUT/ab.txt
