#!/bin/bash
# fused launch in 8-wave workgroups (2 waves per SIMD) that leave registers for the simulator's waves: register budget of 3 / of 2 waves per SIMD, grids
export TMPDIR=/tmp
OUT=gpurun_out/r4w8; mkdir -p $OUT; rm -f $OUT/*.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['march_per_launch_group'][:3], b['in_pipeline_march_per_launch_group'][:3])"; }
echo "base $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
for v in w8p2 w8p3; do
for g in 128 160 192 256; do
echo "$v grid $g $(PN_LIB_PATH=$PWD/pienerf_amd/lib/variants/$v.so PN_FUSED_GRID=$g python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
done
echo "base $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
