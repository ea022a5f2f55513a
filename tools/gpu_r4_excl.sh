#!/bin/bash
# render lanes masked off n CUs that the simulator's launches always find free (the simulator itself unmasked)
export TMPDIR=/tmp
OUT=gpurun_out/r4excl; mkdir -p $OUT
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['in_pipeline_march_per_launch_group'][:3])"; }
echo "base $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "no-substep $(python bench.py --no-extras --no-cpu-baseline --probe no-substep 2>/dev/null | val)" | tee -a $OUT/ab.txt
for n in 8 16 32 48 64; do
echo "excl $n $(PN_PROBE_RENDER_EXCL=$n python bench.py --no-extras --no-cpu-baseline --probe render-excl 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "base $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "excl 32 K20 $(PN_PROBE_RENDER_EXCL=32 python bench.py --no-extras --no-cpu-baseline --probe render-excl --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "base K20 $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
