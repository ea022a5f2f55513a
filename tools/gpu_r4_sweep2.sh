#!/bin/bash
# after the network diet: fused grid / one-lane rounds sweep (three lanes), triangular-solve form of the global step
export TMPDIR=/tmp
OUT=gpurun_out/r4sweep2; mkdir -p $OUT; rm -f $OUT/*.txt
python tools/time_trisolve.py 2>&1 | tail -3 | tee $OUT/trisolve.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['in_pipeline_march_per_launch_group'][:3])"; }
echo "base $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
for g in 96 112 144 160 192; do
echo "grid $g $(PN_FUSED_GRID=$g python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "base $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
for r in 32 48 96; do
echo "rounds $r $(PN_HARNESS_THROUGHPUT=$r python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "form plain $(python bench.py --no-extras --no-cpu-baseline --form plain 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "form whole $(python bench.py --no-extras --no-cpu-baseline --form whole 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "depth 3 $(python bench.py --no-extras --no-cpu-baseline --depth 3 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "base $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
