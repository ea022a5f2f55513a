#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02e
mkdir -p $O
python bench.py --no-cpu-baseline --no-extras --copy-on sim > $O/bench_sim.json 2> $O/e1
python bench.py --no-cpu-baseline --no-extras --copy-on lane > $O/bench_lane.json 2> $O/e2
python bench.py --no-cpu-baseline --no-extras --copy-on sim --depth 3 > $O/bench_sim_d3.json 2> $O/e3
python bench.py --config stress --steps 30 --warmup 4 --no-cpu-baseline --no-extras > $O/bench_stress3.json 2> $O/e6
python bench.py --config stress --steps 30 --warmup 4 --no-cpu-baseline --no-extras --staged-streams 2 > $O/bench_stress2.json 2> $O/e7
for i in 1 2 3 6 7; do tail -c 200 $O/e$i; done
