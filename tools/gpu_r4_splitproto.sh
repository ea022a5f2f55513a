#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4split; mkdir -p $OUT
V=$PWD/pienerf_amd/lib/variants/splitproto.so
python tools/time_net_fixed.py make 2>/dev/null | tee $OUT/net.txt
for i in 1 2; do
python tools/time_net_fixed.py time 2>/dev/null | tee -a $OUT/net.txt
PN_LIB_PATH=$V python tools/time_net_fixed.py time 2>/dev/null | tee -a $OUT/net.txt
done
