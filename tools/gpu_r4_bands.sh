#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4bands; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
PN_FUSED_BANDS=1 timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
PN_FUSED_GRID=128 python tools/fused_clocks.py 2>/dev/null | tail -12 | tee $OUT/clocks_128_rr.txt
PN_FUSED_BANDS=1 PN_FUSED_GRID=128 python tools/fused_clocks.py 2>/dev/null | tail -12 | tee $OUT/clocks_128_bands.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'])"; }
for i in 1 2; do
echo "rr $($B 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "bands $(PN_FUSED_BANDS=1 $B 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
