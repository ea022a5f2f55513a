#!/bin/bash
# fp16 hi/lo split selected as v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16 (1.5 instructions per value, 1 271 instead of 1 439 per tile): tests, A/B against the
# previous library is not possible in one tree, so: bench x2 + the driver's command, the network alone, stress / trex knobs, a fourth lane with more hardware queues
export TMPDIR=/tmp
OUT=gpurun_out/r4mix; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; n=d['network']['all_samples_one_launch']; print(d['value'], d['value_unprimed'], d['verified'], b['render_frame_eager'], b['march_per_launch_group'][:3], b['in_pipeline_march_per_launch_group'][:3], n['launch_ms_fp32'], n['launch_ms_fp16'])"; }
for i in 1 2; do
echo "chair $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "K20 $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "lanes2 $(python bench.py --no-extras --no-cpu-baseline --lanes 2 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "hwq8 lanes3 $(GPU_MAX_HW_QUEUES=8 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "hwq8 lanes4 $(GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-extras --no-cpu-baseline --lanes 4 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "hwq8 lanes5 $(GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-extras --no-cpu-baseline --lanes 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "stress $(python bench.py --config stress --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "stress thr0 $(PN_HARNESS_THROUGHPUT=0 python bench.py --config stress --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "stress thr16 $(PN_HARNESS_THROUGHPUT=16 python bench.py --config stress --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "stress thr32 $(PN_HARNESS_THROUGHPUT=32 python bench.py --config stress --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "trex $(python bench.py --config trex --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
