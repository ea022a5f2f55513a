#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4x; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest.txt
python -m pytest tests/test_gpu_parity.py -q -s -k "nerf_forward_fused_vs_oracle" 2>&1 | grep -i "err\|rel\|abs\|passed" | head -10 | tee $OUT/acc_x.txt
PN_NET_FORM=bf16 python -m pytest tests/test_gpu_parity.py -q -s -k "nerf_forward_fused_vs_oracle" 2>&1 | grep -i "err\|rel\|abs\|passed" | head -10 | tee $OUT/acc_bf16.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; n=d['network']['all_samples_one_launch']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'], b['in_pipeline_march_per_launch_group'], n['launch_ms_fp32'])"; }
for i in 1 2; do
echo "x-form $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "bf16x3 $(PN_NET_FORM=bf16 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "K20 x-form $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "K20 bf16x3 $(PN_NET_FORM=bf16 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
