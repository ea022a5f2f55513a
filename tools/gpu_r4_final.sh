#!/bin/bash
# Round-4 GPU batch on the final tree: test suite, bench lines (chair default / driver-sized / lanes 1, 2 / no D2H / stress / trex), rocprofv3 kernel stats of
# the bench command and of eager frames, HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) for the three workloads in the form the pipeline runs,
# instruction counts per kernel, SQ / TCC / latency passes, fused-launch phase clocks, substep timing, soak.  Output: gpurun_out/r04final/
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04final
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
cd /tmp
P() { cfg=$1; name=$2; shift; shift; rm -rf /tmp/pmc_${cfg}_$name; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_${cfg}_$name -o $name --output-format csv -- python $R/tools/run_frames.py --config $cfg --frames 3 --no-sim --no-counters --form pipeline > /tmp/pmc_${cfg}_$name.log 2>&1 || { echo "pass $cfg $name failed"; return; }; python $R/tools/pmc_summary.py /tmp/pmc_${cfg}_$name k_ > $O/pmc_${cfg}_${name}_per_kernel.txt 2>&1; }
for cfg in chair stress trex; do
  P $cfg fetch FETCH_SIZE
  P $cfg write WRITE_SIZE
  out=$R/profiles/pmc_traffic.json; [ $cfg != chair ] && out=$R/profiles/pmc_traffic_$cfg.json
  (cd $R && python tools/pmc_traffic.py /tmp/pmc_${cfg}_fetch /tmp/pmc_${cfg}_write 3 $out > $O/pmc_traffic_$cfg.txt 2>&1 || echo "traffic $cfg failed")
  cp $out $O/
done
P chair sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
P chair sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH
P chair sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU
P chair lat VmemLatency
P chair tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
(cd $R && python tools/pmc_instr.py /tmp/pmc_chair_sq1 3 k_ > $O/instr_per_kernel_sq1.txt 2>&1; python tools/pmc_instr.py /tmp/pmc_chair_sq2 3 k_ > $O/instr_per_kernel_sq2.txt 2>&1; python tools/pmc_instr.py /tmp/pmc_chair_sq3 3 k_ > $O/instr_per_kernel_sq3.txt 2>&1)
cd $R
python bench.py > $O/bench_chair.json 2> $O/bench_chair.err; tail -c 300 $O/bench_chair.json
python bench.py --steps 20 --warmup 5 > $O/bench_chair_20steps.json 2>/dev/null
python bench.py --config stress > $O/bench_stress.json 2> $O/bench_stress.err
python bench.py --config trex > $O/bench_trex.json 2> $O/bench_trex.err
python bench.py --no-cpu-baseline --no-extras --no-d2h > $O/bench_chair_no_d2h.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --lanes 1 > $O/bench_chair_lanes1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --lanes 2 > $O/bench_chair_lanes2.json 2>/dev/null
cd /tmp
S() { name=$1; shift; rm -rf /tmp/st_$name; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o $name -- "$@" > $O/stats_$name.out 2> /tmp/st_$name.log || echo "stats $name failed"; find /tmp/st_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \; ; }
S chair python $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10
S stress python $R/bench.py --no-cpu-baseline --no-extras --config stress --steps 100 --warmup 10
S trex python $R/bench.py --no-cpu-baseline --no-extras --config trex --steps 100 --warmup 10
S eager python $R/tools/run_frames.py --frames 20 --no-counters --form pipeline
cd $R
python tools/fused_clocks.py 2>/dev/null > $O/fused_clocks.txt; PN_FUSED_GRID=128 python tools/fused_clocks.py 2>/dev/null >> $O/fused_clocks.txt
python tools/time_sim.py > $O/time_sim.txt 2>&1
python tools/time_sim.py --persistent 2>&1 | grep -v amdgpu.ids > $O/time_sim_persistent.txt
timeout 900 python tools/soak.py --frames 3000 > $O/soak_3000.json 2> $O/soak.err; tail -c 400 $O/soak_3000.json
python tools/run_frames.py --frames 2 --form pipeline 2>&1 | grep -E "counters|trips" > $O/trip_records.txt
timeout 600 python tools/time_widened.py > $O/widened_rows.json 2> $O/widened.err
python -c "
import json
for c in ('chair','chair_20steps','stress','trex','chair_lanes1','chair_lanes2','chair_no_d2h'):
    try:
        d=json.load(open('$O/bench_%s.json'%c)); r=d['roofline']; print(c, d['value'], d.get('value_unprimed'), d['ms_per_step'], d.get('verified'), r['frac'], r['traffic'], r.get('launch_ms'), r.get('launch_ms_alone'), d.get('network',{}).get('frac'), d.get('latency_ms_per_step'))
    except Exception as e: print(c, 'ERR', e)
"
ls $O | wc -l
