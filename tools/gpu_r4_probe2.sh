#!/bin/bash
# round 4: CUs reserved for the simulator stream (stream CU masks), and smaller fused grids that leave CUs free
export TMPDIR=/tmp
OUT=gpurun_out/r4probe
mkdir -p $OUT
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['in_pipeline_march_per_launch_group'])"; }
for C in 32 48 64 96; do echo "sim_cus=$C $(PN_PROBE_SIM_CUS=$C $B --probe sim-cus 2>/dev/null | val)" | tee -a $OUT/probe2.txt; done
for C in 32 64; do echo "sim_cus=$C fused_grid=96 $(PN_FUSED_GRID=96 PN_PROBE_SIM_CUS=$C $B --probe sim-cus 2>/dev/null | val)" | tee -a $OUT/probe2.txt; done
for G in 64 80 96 112; do echo "fused_grid=$G $(PN_FUSED_GRID=$G $B 2>/dev/null | val)" | tee -a $OUT/probe2.txt; done
