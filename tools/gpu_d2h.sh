#!/bin/bash
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r03d2h; mkdir -p $O
cd /tmp
python $R/tools/d2h_probe.py
HSA_ENABLE_SDMA=0 python $R/tools/d2h_probe.py
GPU_FORCE_BLIT_COPY_SIZE=0 python $R/tools/d2h_probe.py
GPU_FORCE_BLIT_COPY_SIZE=1000000 python $R/tools/d2h_probe.py
rm -rf /tmp/mc; rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/mc -o d2h --output-format csv -- python $R/tools/d2h_probe.py > /tmp/mc.log 2>&1
ls /tmp/mc/*; for f in /tmp/mc/*/*memory_copy*stats*.csv /tmp/mc/*memory_copy*stats*.csv; do [ -f $f ] && cat $f; done
for f in /tmp/mc/*/*kernel_stats.csv /tmp/mc/*kernel_stats.csv; do [ -f $f ] && head -5 $f | cut -c1-200; done
for f in /tmp/mc/*/*memory_copy_trace.csv /tmp/mc/*memory_copy_trace.csv; do [ -f $f ] && head -8 $f && cp $f $O/; done
