#!/bin/bash
# fused_trip with 4 / 12 window rounds before the 64-lane windows, fused grid 128 / 256
export TMPDIR=/tmp
OUT=gpurun_out/r4trip2; mkdir -p $OUT; rm -f $OUT/*.txt
timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -5 | tee $OUT/pytest_fused.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; c=d['config']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'][:5], b['in_pipeline_march_per_launch_group'][:5])"; }
for v in tr4 tr12; do
for g in 128 256; do
echo "stress $v grid $g $(PN_LIB_PATH=$PWD/pienerf_amd/lib/variants/$v.so PN_FUSED_GRID=$g PN_TRIP_FUSED=1 timeout 600 python bench.py --config stress --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
done
echo "stress per-trip   $(PN_TRIP_FUSED=0 timeout 600 python bench.py --config stress --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "trex tr12 256 $(PN_LIB_PATH=$PWD/pienerf_amd/lib/variants/tr12.so PN_FUSED_GRID=256 PN_TRIP_FUSED=1 timeout 600 python bench.py --config trex --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
