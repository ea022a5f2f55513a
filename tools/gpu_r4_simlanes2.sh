#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4simlanes2; mkdir -p $OUT; rm -f $OUT/*.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['in_pipeline_march_per_launch_group'][:3])"; }
for i in 1 2 3; do
echo "base lanes3             $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "sim-on-lanes lanes4 g128 $(PN_FUSED_GRID=128 python bench.py --no-extras --no-cpu-baseline --sim-on-lanes --lanes 4 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "sim-on-lanes lanes4 g112 $(PN_FUSED_GRID=112 python bench.py --no-extras --no-cpu-baseline --sim-on-lanes --lanes 4 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "sim-on-lanes lanes4 g144 $(PN_FUSED_GRID=144 python bench.py --no-extras --no-cpu-baseline --sim-on-lanes --lanes 4 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "sim-on-lanes lanes4 g128 depth1 $(PN_FUSED_GRID=128 python bench.py --no-extras --no-cpu-baseline --sim-on-lanes --lanes 4 --depth 1 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "sim-on-lanes lanes4 g128 plain $(PN_FUSED_GRID=128 python bench.py --no-extras --no-cpu-baseline --sim-on-lanes --lanes 4 --form plain 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "sim-on-lanes lanes4 g128 whole $(PN_FUSED_GRID=128 python bench.py --no-extras --no-cpu-baseline --sim-on-lanes --lanes 4 --form whole 2>/dev/null | val)" | tee -a $OUT/ab.txt
for i in 1 2; do
echo "base K20                $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "sim-on-lanes lanes4 K20 $(PN_FUSED_GRID=128 python bench.py --no-extras --no-cpu-baseline --sim-on-lanes --lanes 4 --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
