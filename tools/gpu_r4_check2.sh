#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4check2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['stepforward_alone'], b['stepforward_persistent_alone'], b['in_pipeline_march_per_launch_group'])"; }
for i in 1 2; do echo "default $($B 2>/dev/null | val)" | tee -a $OUT/lines.txt; done
echo "mv256 $(PN_SIM_MV_WG=256 $B 2>/dev/null | val)" | tee -a $OUT/lines.txt
echo "lanes2 $($B --lanes 2 2>/dev/null | val)" | tee -a $OUT/lines.txt
echo "20 steps $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/lines.txt
