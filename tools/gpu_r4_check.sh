#!/bin/bash
# round 4: test suite + the bench lines of the default configuration
export TMPDIR=/tmp
OUT=gpurun_out/r4check
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['value_unprimed'], d['config']['frames_continued_past_captured_trips'], d['breakdown_ms']['render_frame_eager'])"; }
for L in 3 2 1; do echo "default lanes=$L $($B --lanes $L 2>$OUT/err_$L.txt | val)" | tee -a $OUT/lines.txt; done
echo "20 steps $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/lines.txt
echo "stress $(python bench.py --no-extras --no-cpu-baseline --config stress 2>/dev/null | val)" | tee -a $OUT/lines.txt
echo "trex $(python bench.py --no-extras --no-cpu-baseline --config trex 2>/dev/null | val)" | tee -a $OUT/lines.txt
