#!/bin/bash
# fused trips for frames in ray batches (configs[4]): tests, then stress bench A/B
export TMPDIR=/tmp
OUT=gpurun_out/r4groups; mkdir -p $OUT; rm -f $OUT/*.txt
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -25 | tee $OUT/pytest_fused.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "batch or stress or staged" 2>&1 | tail -8 | tee $OUT/pytest_groups.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; c=d['config']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b.get('fused_from_trip'), c['trips_per_frame'], c['frames_continued_past_captured_trips'], b['in_pipeline_march_per_launch_group'][:6])"; }
for m in 0 1; do
echo "stress fused margin $m $(PN_GROUP_FUSE_MARGIN=$m python bench.py --config stress --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "stress per-trip $(PN_GROUP_FUSE=0 python bench.py --config stress --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "stress fused lanes2 $(python bench.py --config stress --no-extras --no-cpu-baseline --lanes 2 2>/dev/null | val)" | tee -a $OUT/ab.txt
