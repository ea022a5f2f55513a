#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r4clk
python tools/fused_clocks.py 2>/dev/null | tee gpurun_out/r4clk/clocks_256.txt
PN_FUSED_GRID=128 python tools/fused_clocks.py 2>/dev/null | tee gpurun_out/r4clk/clocks_128.txt
