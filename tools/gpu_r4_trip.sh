#!/bin/bash
# pn_render_opts.fused_trip: a later trip's march + network as one launch (pn_trips_fused.h MODE 3): tests, then stress / trex / chair A/B through PN_TRIP_FUSED
export TMPDIR=/tmp
OUT=gpurun_out/r4trip; mkdir -p $OUT; rm -f $OUT/*.txt
timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -30 | tee $OUT/pytest_fused.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; c=d['config']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], c['trips_per_frame'], c['frames_continued_past_captured_trips'], b['march_per_launch_group'][:5], b['in_pipeline_march_per_launch_group'][:5])"; }
for i in 1 2; do
echo "stress trip-fused $(PN_TRIP_FUSED=1 timeout 600 python bench.py --config stress --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "stress per-trip   $(PN_TRIP_FUSED=0 timeout 600 python bench.py --config stress --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "trex trip-fused $(PN_TRIP_FUSED=1 timeout 600 python bench.py --config trex --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "trex per-trip   $(PN_TRIP_FUSED=0 timeout 600 python bench.py --config trex --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "chair trip-fused $(PN_TRIP_FUSED=1 timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "chair per-trip   $(PN_TRIP_FUSED=0 timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
