#!/bin/bash
# byte-offset form of the fp32 encoder (one v_xad_u32 per corner, 1 108 vector instructions per tile) + the gather calibration
export TMPDIR=/tmp
OUT=gpurun_out/r4bytes; mkdir -p $OUT
tools/bin/calib_gather 2>&1 | tee $OUT/calib_gather.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; n=d['network']['all_samples_one_launch']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'][:3], b['in_pipeline_march_per_launch_group'][:3], n['launch_ms_fp32'], n['launch_ms_fp16'])"; }
for i in 1 2; do
echo "chair $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "K20 $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "trex $(python bench.py --config trex --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
