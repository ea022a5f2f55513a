#!/bin/bash
# Round-2 final GPU batch: test suite, bench lines (chair / stress / trex / sigma-gain sweep), rocprofv3 kernel stats of the same commands,
# SQ / TCC / TCP counter passes and HBM traffic of the shipped kernels, calibration microbenchmarks, soak.  Output: gpurun_out/r02final/
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02final
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python bench.py > $O/bench_chair.json 2> $O/bench_chair.err; tail -c 300 $O/bench_chair.json
python bench.py --config stress > $O/bench_stress.json 2> $O/bench_stress.err
python bench.py --config stress --whole-frame --no-cpu-baseline --no-extras > $O/bench_stress_whole_frame.json 2>/dev/null
python bench.py --config trex > $O/bench_trex.json 2> $O/bench_trex.err
for g in 0.3 3.0; do python bench.py --no-cpu-baseline --no-extras --sigma-gain $g > $O/bench_chair_sigma_gain_$g.json 2>/dev/null; done
python bench.py --no-cpu-baseline --no-extras --no-d2h > $O/bench_chair_no_d2h.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --lanes 1 > $O/bench_chair_lanes1.json 2>/dev/null
cd /tmp
S() { name=$1; shift; rm -rf /tmp/st_$name; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o $name -- "$@" > $O/stats_$name.bench.json 2> /tmp/st_$name.log || echo "stats $name failed"; cp /tmp/st_$name/*kernel_stats.csv $O/${name}_kernel_stats.csv 2>/dev/null || find /tmp/st_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \; ; }
S chair python $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10
S stress python $R/bench.py --no-cpu-baseline --no-extras --config stress --steps 3 --warmup 1
S trex python $R/bench.py --no-cpu-baseline --no-extras --config trex --steps 100 --warmup 10
S eager python $R/tools/run_frames.py --frames 5
P() { name=$1; shift; rm -rf /tmp/pmc_$name; timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o $name --output-format csv -- python $R/tools/run_frames.py --frames 3 --no-sim --no-counters > /tmp/pmc_$name.log 2>&1 || { echo "pass $name failed"; return; }; python $R/tools/pmc_summary.py /tmp/pmc_$name k_ > $O/pmc_${name}_per_kernel.txt 2>&1; }
P sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
P sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH
P sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU
P lat VmemLatency
P tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
P tcp TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum
P fetch FETCH_SIZE
P write WRITE_SIZE
cd $R
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write 3 $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1 || echo "traffic failed"
python tools/pmc_dispatch.py /tmp/pmc_sq1 "k_march" 0 40 > $O/pmc_sq1_march_dispatches.txt 2>&1
tools/bin/calib_clock > $O/calib_clock.txt 2>&1
tools/bin/calib_tcp > $O/calib_tcp.txt 2>&1
tools/bin/calib_atomic > $O/calib_atomic.txt 2>&1
timeout 600 python tools/soak.py --frames 1500 > $O/soak_1500.json 2> $O/soak.err; tail -c 400 $O/soak_1500.json
python tools/run_frames.py --frames 2 2>&1 | grep -E "counters|trips" > $O/trip_records.txt
ls -la $O | head -60
