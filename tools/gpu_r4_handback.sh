#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4hb; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3 | tee $OUT/pytest.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'])"; }
for H in 0 4 2 7 0 4; do echo "handback=$H $(PN_FUSED_HANDBACK=$H python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt; done
for H in 0 4; do echo "lanes2 handback=$H $(PN_FUSED_HANDBACK=$H python bench.py --no-extras --no-cpu-baseline --lanes 2 2>/dev/null | val)" | tee -a $OUT/ab.txt; done
for H in 0 4; do echo "K20 handback=$H $(PN_FUSED_HANDBACK=$H python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt; done
