#!/bin/bash
# per-layer power-of-two scales folded into the fp16 hi/lo weight image: netform tests first, then the whole GPU suite, then bench x2 + driver's command
export TMPDIR=/tmp
OUT=gpurun_out/r4scales; mkdir -p $OUT
python -m pytest tests/test_gpu_netform.py -m gpu -q 2>&1 | tail -30 | tee $OUT/netform.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; n=d['network']['all_samples_one_launch']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'][:3], b['in_pipeline_march_per_launch_group'][:3], n['launch_ms_fp32'], n['launch_ms_fp16'])"; }
for i in 1 2; do
echo "chair $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "K20 $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
