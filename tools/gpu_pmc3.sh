#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/$1; PAT=$2
mkdir -p $O
R=$PWD
cd /tmp
P() { name=$1; shift; rm -rf /tmp/pmc_$name; timeout 150 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o $name --output-format csv -- python $R/tools/run_frames.py --frames 2 --no-sim > /tmp/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 /tmp/pmc_$name.log; return; }; python $R/tools/pmc_dispatch.py /tmp/pmc_$name "$PAT" 8 6 | tee $O/pmc_$name.txt; }
P f VmemLatency
P g LdsLatency SmemLatency
P h InstrFetchLatency MeanOccupancyPerActiveCU
P i SQC_ICACHE_MISSES SQC_ICACHE_REQ SQC_DCACHE_MISSES SQC_DCACHE_REQ SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_INSTS_LDS
P j SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
