#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4fold; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -12 | tee $OUT/pytest.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'])"; }
for i in 1 2; do
echo "fold $($B 2>$OUT/err.txt | val)" | tee -a $OUT/ab.txt
echo "no fold $(PN_FUSED_FOLD=0 $B 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
tail -3 $OUT/err.txt
echo "fold 20 steps $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "no fold 20 steps $(PN_FUSED_FOLD=0 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "fold lanes 1 $($B --lanes 1 2>/dev/null | val)" | tee -a $OUT/ab.txt
