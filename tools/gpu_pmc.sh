#!/bin/bash
# usage: gpu_pmc.sh <outdir> <pattern> -- per-dispatch SQ counters of the kernels matching <pattern> in an eager 3-frame run
export TMPDIR=/tmp
O=$PWD/gpurun_out/$1; PAT=$2
mkdir -p $O
R=$PWD
cd /tmp
P() { name=$1; shift; rm -rf /tmp/pmc_$name; rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o $name --output-format csv -- python $R/tools/run_frames.py --frames 2 --no-sim > /tmp/pmc_$name.log 2>&1 || echo "pass $name failed"; python $R/tools/pmc_dispatch.py /tmp/pmc_$name "$PAT" 8 8 | tee $O/pmc_$name.txt; }
P a SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES
P b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY
