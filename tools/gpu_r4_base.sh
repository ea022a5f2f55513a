#!/bin/bash
# round 4 baseline on the re-entry tree: test suite, bench lines (chair default / driver-sized / lanes 1-2 / classic trips / stress / trex), kernel stats
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4base
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python bench.py > $O/bench_chair.json 2> $O/bench_chair.err; tail -c 300 $O/bench_chair.json
python bench.py --steps 20 --warmup 5 > $O/bench_chair_20steps.json 2>/dev/null
for L in 1 2; do python bench.py --no-cpu-baseline --no-extras --lanes $L > $O/bench_chair_lanes$L.json 2>/dev/null; done
PN_FUSED=0 python bench.py --no-cpu-baseline --no-extras > $O/bench_chair_classic.json 2>/dev/null
python bench.py --config stress --no-cpu-baseline > $O/bench_stress.json 2> $O/bench_stress.err
python bench.py --config trex --no-cpu-baseline > $O/bench_trex.json 2> $O/bench_trex.err
cd /tmp
S() { name=$1; shift; rm -rf /tmp/st_$name; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o $name -- "$@" > $O/stats_$name.out 2> /tmp/st_$name.log || echo "stats $name failed"; find /tmp/st_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \; ; }
S chair python $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10
S eager python $R/tools/run_frames.py --frames 20 --no-counters
cd $R
python -c "
import json
for c in ('chair','chair_20steps','chair_lanes1','chair_lanes2','chair_classic','stress','trex'):
    try:
        d=json.load(open('$O/bench_%s.json'%c)); print(c, d['value'], d.get('value_unprimed'), d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('ms_per_frame'), d.get('network',{}).get('frac'), d.get('latency_ms_per_step'))
    except Exception as e: print(c, 'ERR', e)
"
head -12 $O/chair_kernel_stats.csv | cut -c1-150
