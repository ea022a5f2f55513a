#!/bin/bash
# EXPERIMENT PN_ENC_PAIR: x / x+1 corners of the fp32 encoder as one 16-byte load where they are neighbours
export TMPDIR=/tmp
OUT=gpurun_out/r4pair; mkdir -p $OUT; rm -f $OUT/*.txt
V=$PWD/pienerf_amd/lib/variants/pair.so
PN_LIB_PATH=$V timeout 600 python -m pytest tests/test_gpu_netform.py tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest.txt
python tools/time_net_fixed.py make 2>&1 | tail -1
for i in 1 2; do
python tools/time_net_fixed.py time 2>&1 | tail -1 | tee -a $OUT/net.txt
PN_LIB_PATH=$V python tools/time_net_fixed.py time 2>&1 | tail -1 | tee -a $OUT/net.txt
done
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; n=d['network']['all_samples_one_launch']; print(d['value'], d['value_unprimed'], d['verified'], b['march_per_launch_group'][:3], b['in_pipeline_march_per_launch_group'][:3], n['launch_ms_fp32'])"; }
for i in 1 2; do
echo "base $(python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
echo "pair $(PN_LIB_PATH=$V python bench.py --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $OUT/ab.txt
done
