#!/bin/bash
# no simulator stream: substep g on render lane g % lanes (frames.FramePipeline.sim_on_lanes) — the freed hardware queue as a fourth render lane
export TMPDIR=/tmp
OUT=gpurun_out/r4simlanes; mkdir -p $OUT; rm -f $OUT/*.txt
timeout 600 python -m pytest tests/test_frames_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['in_pipeline_march_per_launch_group'][:3])"; }
echo "base lanes3            $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "sim-on-lanes lanes3    $(python bench.py --no-extras --no-cpu-baseline --sim-on-lanes 2>/dev/null | val)" | tee -a $OUT/ab.txt
for g in 96 128 160; do
echo "sim-on-lanes lanes4 g$g $(PN_FUSED_GRID=$g python bench.py --no-extras --no-cpu-baseline --sim-on-lanes --lanes 4 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "sim-on-lanes lanes5 g96 $(PN_FUSED_GRID=96 timeout 300 python bench.py --no-extras --no-cpu-baseline --sim-on-lanes --lanes 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "base K20               $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "sim-on-lanes lanes4 K20 $(PN_FUSED_GRID=128 python bench.py --no-extras --no-cpu-baseline --sim-on-lanes --lanes 4 --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "sim-on-lanes lanes3 K20 $(python bench.py --no-extras --no-cpu-baseline --sim-on-lanes --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "base lanes3            $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
