#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4early; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $OUT/pytest.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'])"; }
for i in 1 2; do
echo "K200 early $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "K200 finish-launch $(PN_EARLY_FINISH=0 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
for i in 1 2; do
echo "K20 early $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "K20 finish-launch $(PN_EARLY_FINISH=0 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "lanes2 early $(python bench.py --no-extras --no-cpu-baseline --lanes 2 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "lanes2 finish-launch $(PN_EARLY_FINISH=0 python bench.py --no-extras --no-cpu-baseline --lanes 2 2>/dev/null | val)" | tee -a $OUT/ab.txt
