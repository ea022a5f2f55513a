#!/bin/bash
# calc_elastic + chunked gather as one launch (k_elastic_gather): sim tests, substep alone, pipeline A/B via PN_SIM_FUSE_EG
export TMPDIR=/tmp
OUT=gpurun_out/r4eg; mkdir -p $OUT; rm -f $OUT/*.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_persistent.py tests/test_gpu_simpin.py tests/test_gpu_edges.py -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest.txt
for w in 1 2 8; do echo "EG waves $w: $(PN_SIM_EG_WAVES=$w timeout 300 python tools/time_sim.py 2>&1 | tail -1)" | tee -a $OUT/time_sim.txt; done
echo "two launches: $(PN_SIM_FUSE_EG=0 timeout 300 python tools/time_sim.py 2>&1 | tail -1)" | tee -a $OUT/time_sim.txt
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], b['stepforward_alone'], b['in_pipeline_march_per_launch_group'][:3])"; }
for i in 1 2; do
echo "EG     $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "two    $(PN_SIM_FUSE_EG=0 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
echo "EG K20  $(python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "two K20 $(PN_SIM_FUSE_EG=0 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/ab.txt
