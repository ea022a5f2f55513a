#!/bin/bash
# One parametrised GPU batch script (replaces the per-experiment gpu_*.sh run logs of rounds 1-4; their outputs live in profiles/, the scripts in git history).
#   gpurun --timeout 1800 -- 'bash tools/gpu_run.sh <out-name> <job> [<job> ...]'
# Jobs (each writes under gpurun_out/<out-name>/):
#   tests[:expr]        pytest -m gpu (-k expr)
#   bench[:cfg[:args]]  python bench.py --config cfg args   (cfg chair|trex|stress; args '+'-separated, e.g. bench:chair:--steps+20+--warmup+5)
#   stats[:cfg]         rocprofv3 --kernel-trace --stats of a 100-step bench run -> <cfg>_kernel_stats.csv
#   eager[:cfg]         rocprofv3 --kernel-trace --stats of 20 eager frames in the pipeline's launch forms
#   pmc:cfg:name:C1+C2  one --pmc pass (counters '+'-separated) over 3 pipeline-form frames -> pmc_<cfg>_<name>_per_kernel.txt
#   traffic[:cfg]       FETCH_SIZE and WRITE_SIZE passes (separate) -> profiles/pmc_traffic[_cfg].json
#   sim                 tools/time_sim.py (launch form and persistent form)
#   clocks              tools/fused_clocks.py (phase clocks of the fused launch)
#   ab:ENV=a,b[:cfg]    alternating A/B of one environment knob over bench.py (3 rounds)
#   variant:NAME:cfg    bench.py with PN_LIB_PATH=pienerf_amd/lib/variants/NAME.so (tools/build_variant.py)
#   py:script[:args]    python tools/<script>.py args ('+'-separated)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$1; shift
mkdir -p $O
form() { [ "$1" = chair ] && echo fold || echo pipeline; }   # the launch set bench.py times: on the chair the first trip is folded into the fused launch (k_trips_fused<.., 2>)
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('ms_per_step'), d.get('verified'), d['roofline']['frac'] if d.get('roofline') else None)" 2>/dev/null || echo ERR; }
for job in "$@"; do
  IFS=: read -r kind a b c <<< "$job"
  case $kind in
    tests) if [ -n "$a" ]; then python -m pytest tests -m gpu -q -x -k "$a" > $O/pytest_gpu_$(echo $a | tr ' ' _).txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed|^E  .*(assert|Error)" $O/pytest_gpu_$(echo $a | tr ' ' _).txt | cut -c1-400 | tail -20; else python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed|^E  .*(assert|Error)" $O/pytest_gpu.txt | cut -c1-400 | tail -30; fi ;;
    bench) cfg=${a:-chair}; args=$(echo "$b" | tr + ' '); tag=$(echo "$b" | tr -c 'a-zA-Z0-9\n' _)
           python bench.py --config $cfg $args > $O/bench_${cfg}${tag}.json 2> $O/bench_${cfg}${tag}.err; echo "bench $cfg $args: $(val < $O/bench_${cfg}${tag}.json)" | tee -a $O/summary.txt ;;
    stats) cfg=${a:-chair}; (cd /tmp && rm -rf /tmp/st_$cfg && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$cfg -o $cfg -- python $R/bench.py --no-cpu-baseline --no-extras --config $cfg --steps 100 --warmup 10 > $O/stats_$cfg.out 2> /tmp/st_$cfg.log || echo "stats $cfg failed"; find /tmp/st_$cfg -name "*kernel_stats.csv" -exec cp {} $O/${cfg}_kernel_stats.csv \; ); head -12 $O/${cfg}_kernel_stats.csv ;;
    eager) cfg=${a:-chair}; (cd /tmp && rm -rf /tmp/st_eager_$cfg && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_eager_$cfg -o eager -- python $R/tools/run_frames.py --config $cfg --frames 20 --no-counters --form $(form $cfg) > $O/eager_$cfg.out 2> /tmp/st_eager_$cfg.log || echo "eager $cfg failed"; find /tmp/st_eager_$cfg -name "*kernel_stats.csv" -exec cp {} $O/eager_${cfg}_kernel_stats.csv \; ); head -12 $O/eager_${cfg}_kernel_stats.csv ;;
    pmc) cfg=$a; name=$b; ctrs=$(echo "$c" | tr + ' ')
         (cd /tmp && rm -rf /tmp/pmc_${cfg}_$name && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/pmc_${cfg}_$name -o $name --output-format csv -- python $R/tools/run_frames.py --config $cfg --frames 3 --no-sim --no-counters --form $(form $cfg) > /tmp/pmc_${cfg}_$name.log 2>&1 || echo "pass $cfg $name failed"; python $R/tools/pmc_summary.py /tmp/pmc_${cfg}_$name k_ > $O/pmc_${cfg}_${name}_per_kernel.txt 2>&1) ;;
    traffic) cfg=${a:-chair}
         for ctr in FETCH_SIZE WRITE_SIZE; do n=$(echo $ctr | tr A-Z a-z | cut -d_ -f1); (cd /tmp && rm -rf /tmp/pmc_${cfg}_$n && timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_${cfg}_$n -o $n --output-format csv -- python $R/tools/run_frames.py --config $cfg --frames 3 --no-sim --no-counters --form $(form $cfg) > /tmp/pmc_${cfg}_$n.log 2>&1 || echo "pass $cfg $n failed"); done
         out=$R/profiles/pmc_traffic.json; [ $cfg != chair ] && out=$R/profiles/pmc_traffic_$cfg.json
         python tools/pmc_traffic.py /tmp/pmc_${cfg}_fetch /tmp/pmc_${cfg}_write 3 $out > $O/pmc_traffic_$cfg.txt 2>&1 || echo "traffic $cfg failed"; cp $out $O/ ;;
    sim) python tools/time_sim.py 2>&1 | grep -v amdgpu.ids | tee $O/time_sim.txt; python tools/time_sim.py --persistent 2>&1 | grep -v amdgpu.ids | tee $O/time_sim_persistent.txt ;;
    clocks) python tools/fused_clocks.py 2>/dev/null | tee $O/fused_clocks.txt ;;
    ab) knob=${a%%=*}; vals=${a#*=}; cfg=${b:-chair}
        for r in 1 2 3; do for v in $(echo $vals | tr , ' '); do echo "$knob=$v $cfg round $r: $(env $knob=$v python bench.py --config $cfg --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $O/ab_$knob.txt; done; done ;;
    variant) echo "variant $a ${b:-chair}: $(PN_LIB_PATH=$R/pienerf_amd/lib/variants/$a.so python bench.py --config ${b:-chair} --no-extras --no-cpu-baseline 2>/dev/null | val)" | tee -a $O/variants.txt ;;
    py) args=$(echo "$b" | tr + ' '); python tools/$a.py $args 2>&1 | grep -v amdgpu.ids | tee $O/py_$a.txt | tail -40 ;;
    *) echo "unknown job $job" ;;
  esac
done
ls $O
