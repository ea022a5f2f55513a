#!/bin/bash
# Round-2 GPU batch A: full GPU test suite, hardware repro tools, network timings, a bench line, SQ/TCC counter passes of the shipped kernels.
set -x
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt | tail -5
tools/bin/repro_mfma_overlap > $O/repro_mfma_overlap.json 2>&1
timeout 600 tools/bin/repro_pk_mfma 400000 > $O/repro_pk_mfma.json 2>&1
cat $O/repro_mfma_overlap.json $O/repro_pk_mfma.json
python tools/time_net.py > $O/time_net.txt 2>&1; tail -3 $O/time_net.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_lanes3.json 2> $O/bench_lanes3.err; tail -c 600 $O/bench_lanes3.json
rocprofv3 -L > $O/counters_list.txt 2>&1
cd /tmp
R=/root/repo
P() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d $R/$O/pmc_$name -o $name --output-format csv -- python $R/tools/run_frames.py --frames 3 --no-sim > $R/$O/pmc_$name.log 2>&1 || echo "pass $name failed"; }
P sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
P sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
P sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU
P tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
P grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $R
for n in sq1 sq2 sq3 tcc grbm; do python tools/pmc_summary.py $O/pmc_$n k_ > $O/pmc_${n}_summary.txt 2>&1; done
ls $O
