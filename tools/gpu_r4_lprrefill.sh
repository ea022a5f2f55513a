#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4refill; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_ref.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py -x -q 2>&1 | tail -4 | tee $OUT/pytest.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'])"; }
for W in 0 1 2 4 0 2; do echo "refill workers=$W $(PN_LPR_REFILL=$W $B 2>/dev/null | val)" | tee -a $OUT/ab.txt; done
for W in 0 2; do echo "20 steps workers=$W $(PN_LPR_REFILL=$W python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt; done
for W in 0 2; do echo "stress workers=$W $(PN_LPR_REFILL=$W $B --config stress 2>/dev/null | val)" | tee -a $OUT/ab.txt; done
for W in 0 2; do echo "trex workers=$W $(PN_LPR_REFILL=$W $B --config trex 2>/dev/null | val)" | tee -a $OUT/ab.txt; done
