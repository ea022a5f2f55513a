#!/bin/bash
# Re-measures what depends on the exact kernel sources: HBM traffic (stamped with the source hash) and the three bench lines.  Output: gpurun_out/r02final/
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02final; mkdir -p $O
cd /tmp
P() { name=$1; shift; rm -rf /tmp/pmc_$name; timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o $name --output-format csv -- python $R/tools/run_frames.py --frames 3 --no-sim --no-counters > /tmp/pmc_$name.log 2>&1 || { echo "pass $name failed"; return; }; python $R/tools/pmc_summary.py /tmp/pmc_$name k_ > $O/pmc_${name}_per_kernel.txt 2>&1; }
P fetch FETCH_SIZE
P write WRITE_SIZE
cd $R
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write 3 $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1 || echo "traffic failed"
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $O/bench_chair.json 2> $O/bench_chair.err
python bench.py --config stress > $O/bench_stress.json 2> $O/bench_stress.err
python bench.py --config trex > $O/bench_trex.json 2> $O/bench_trex.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_chair_20steps.json 2>/dev/null
python -m pytest tests -m gpu -q 2>&1 | tail -2 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "
import json
for c in ('chair','stress','trex','chair_20steps'):
    d=json.load(open('$O/bench_%s.json'%c)); print(c, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('ms_per_frame'), d['breakdown_ms']['stepforward_alone'])
"
