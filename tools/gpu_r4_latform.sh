#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4lat; mkdir -p $OUT
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'])"; }
for T in 64 0 64 0; do echo "throughput=$T $(PN_HARNESS_THROUGHPUT=$T python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt; done
for R in 1 3 4; do echo "latency form tail_rounds=$R $(PN_TAIL_ROUNDS=$R PN_HARNESS_THROUGHPUT=0 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt; done
echo "K20 throughput=0 $(PN_HARNESS_THROUGHPUT=0 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "K20 throughput=64 $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
