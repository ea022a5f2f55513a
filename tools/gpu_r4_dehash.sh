#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4dehash; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dehash.py tests/test_gpu_parity.py tests/test_gpu_half.py -x -q 2>&1 | tail -8 | tee $OUT/pytest.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; n=d['network']['all_samples_one_launch']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'], n['launch_ms_fp32'])"; }
for MB in 0 32 100 0 32 100 300; do echo "dehash_mb=$MB $(PN_DEHASH_MB=$MB $B 2>/dev/null | val)" | tee -a $OUT/ab.txt; done
