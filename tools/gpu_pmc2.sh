#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/$1; PAT=$2
mkdir -p $O
R=$PWD
cd /tmp
P() { name=$1; shift; rm -rf /tmp/pmc_$name; rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o $name --output-format csv -- python $R/tools/run_frames.py --frames 2 --no-sim > /tmp/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 /tmp/pmc_$name.log; }; python $R/tools/pmc_dispatch.py /tmp/pmc_$name "$PAT" 8 6 | tee $O/pmc_$name.txt; }
P c TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
P d TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
P e TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
