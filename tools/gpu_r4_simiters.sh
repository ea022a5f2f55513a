#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4simit; mkdir -p $OUT
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['stepforward_alone'], b['in_pipeline_march_per_launch_group'])"; }
for I in 2 5 10 15 20; do echo "sim_iters=$I $(PN_PROBE_SIM_ITERS=$I $B 2>/dev/null | val)" | tee -a $OUT/sweep.txt; done
