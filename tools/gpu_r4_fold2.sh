#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4fold; mkdir -p $OUT
PN_FUSED_GRID=128 PN_FUSED_FOLD=0 python tools/fused_clocks.py 2>/dev/null | tail -13 | tee $OUT/clocks_nofold.txt
PN_FUSED_GRID=128 PN_FUSED_FOLD=1 python tools/fused_clocks.py 2>/dev/null | tail -13 | tee $OUT/clocks_fold.txt
