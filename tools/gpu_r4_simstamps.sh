#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4simstamps; mkdir -p $O
V=$PWD/pienerf_amd/lib/variants
run() { echo "== $*" | tee -a $O/stamps2.txt; env "$@" python tools/sim_stamps.py 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps2.txt; }
run PN_LIB_PATH=$V/simstamps.so PN_SIM_MV_WG=256
run PN_LIB_PATH=$V/simstamps.so PN_SIM_MV_WG=64
run PN_LIB_PATH=$V/simstamps_gch64.so PN_SIM_MV_WG=64
run PN_LIB_PATH=$V/simstamps_gch32.so PN_SIM_MV_WG=64
