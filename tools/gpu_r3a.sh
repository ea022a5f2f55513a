#!/bin/bash
# round 3, batch A: correctness of the new prologue + eager kernel stats + latency bench
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r03a; mkdir -p $O
python -m pytest tests -m gpu -q -x --timeout 1200 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
rm -rf /tmp/ks; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o eager --output-format csv -- python $R/tools/run_frames.py --frames 20 --no-counters > /tmp/ks.log 2>&1
cp /tmp/ks/eager_kernel_stats.csv $O/eager_kernel_stats.csv 2>/dev/null || find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/eager_kernel_stats.csv \;
cd $R
head -30 $O/eager_kernel_stats.csv | cut -c1-150
python bench.py --lanes 1 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_lanes1.json 2> $O/bench_lanes1.err
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_chair.json 2> $O/bench_chair.err
python -c "
import json
for c in ('lanes1','chair'):
    d=json.load(open('$O/bench_%s.json'%c)); print(c, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('ms_per_frame'), d.get('latency_ms_per_step'), d.get('breakdown_ms'))
"
