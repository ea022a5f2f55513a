#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4trex; mkdir -p $OUT
B="python bench.py --config trex --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['in_pipeline_march_per_launch_group'])"; }
for i in 1 2; do
echo "one-wg scan $($B 2>/dev/null | val)" | tee -a $OUT/scan.txt
echo "tiled scan $(PN_TILED_SCAN=1 $B 2>/dev/null | val)" | tee -a $OUT/scan.txt
done
for G in 64 96 256; do echo "fused_grid=$G $(PN_FUSED_GRID=$G $B 2>/dev/null | val)" | tee -a $OUT/scan.txt; done
