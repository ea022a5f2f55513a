#!/bin/bash
# quick: a few parity tests + eager kernel stats of the chair frame
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r03b; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 1200 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
rm -rf /tmp/ks; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o eager --output-format csv -- python $R/tools/run_frames.py --frames 20 --no-counters --no-sim > /tmp/ks.log 2>&1
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/eager_kernel_stats.csv \;
cd $R
grep -E "k_|copyBuffer" $O/eager_kernel_stats.csv | cut -d, -f1-4,6-7 | sed 's/(.*)"/"/' | head -30
