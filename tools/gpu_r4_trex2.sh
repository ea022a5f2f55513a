#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4trex; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_fused.py tests/test_gpu_ref.py -x -q 2>&1 | tail -3
PN_SKIP_COARSE=0 timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -k "trex or cut" 2>&1 | tail -2
B="python bench.py --config trex --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'])"; }
for i in 1 2; do
echo "coarse filter $($B 2>/dev/null | val)" | tee -a $OUT/coarse.txt
echo "no filter $(PN_SKIP_COARSE=0 $B 2>/dev/null | val)" | tee -a $OUT/coarse.txt
done
