#!/bin/bash
# round 4: the persistent substep inside the render pipeline
export TMPDIR=/tmp
OUT=gpurun_out/r4probe
mkdir -p $OUT
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['in_pipeline_march_per_launch_group'])"; }
echo "coop $(PN_SIM_COOP=1 $B 2>$OUT/err_coop.txt | val)" | tee -a $OUT/probe3.txt
tail -3 $OUT/err_coop.txt
echo "coop grid=96 $(PN_SIM_COOP=1 PN_FUSED_GRID=96 $B 2>/dev/null | val)" | tee -a $OUT/probe3.txt
