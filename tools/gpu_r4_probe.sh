#!/bin/bash
# round 4: is the pipeline bound by its render lanes or by the simulator stream?
export TMPDIR=/tmp
OUT=gpurun_out/r4probe
mkdir -p $OUT
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['in_pipeline_march_per_launch_group'])"; }
for P in none no-substep sim-priority sim-cus; do echo "probe=$P $($B --probe $P 2>$OUT/err_$P.txt | val)" | tee -a $OUT/probe.txt; done
for P in none no-substep; do echo "lpr=8 probe=$P $(PN_MARCH_LPR=8 $B --probe $P 2>/dev/null | val)" | tee -a $OUT/probe.txt; done
for P in none no-substep; do echo "lanes=2 probe=$P $($B --lanes 2 --probe $P 2>/dev/null | val)" | tee -a $OUT/probe.txt; done
echo "lanes=4 no-substep $($B --lanes 4 --probe no-substep 2>/dev/null | val)" | tee -a $OUT/probe.txt
tail -2 $OUT/err_no-substep.txt
