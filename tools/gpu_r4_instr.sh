#!/bin/bash
# round 4: where a frame's issue slots go — per-kernel instruction counts of the pipeline-form frame (and its substep), rocprofv3 PMC passes
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4instr
mkdir -p $O
cd /tmp
P() { name=$1; shift; rm -rf /tmp/pi_$name; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pi_$name -o $name --output-format csv -- python $R/tools/run_frames.py --frames 4 --no-counters --form pipeline > /tmp/pi_$name.log 2>&1 || { echo "pass $name failed"; tail -3 /tmp/pi_$name.log; return; }; python $R/tools/pmc_instr.py /tmp/pi_$name 4 > $O/instr_$name.txt 2>&1; }
P a SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
P b SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM
cat $O/instr_a.txt | cut -c1-250 | head -40
