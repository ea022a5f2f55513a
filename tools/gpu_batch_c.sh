#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -40 > $O/pytest_fullsize.txt; tail -5 $O/pytest_fullsize.txt
python bench.py > $O/bench_chair.json 2> $O/bench_chair.err; tail -c 400 $O/bench_chair.json; tail -3 $O/bench_chair.err
python bench.py --config stress --steps 20 --warmup 3 > $O/bench_stress.json 2> $O/bench_stress.err; tail -c 300 $O/bench_stress.json; tail -3 $O/bench_stress.err
python bench.py --config trex --steps 100 --warmup 10 > $O/bench_trex.json 2> $O/bench_trex.err; tail -c 300 $O/bench_trex.json; tail -3 $O/bench_trex.err
for g in 0.3 3.0; do python bench.py --sigma-gain $g --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_chair_gain$g.json 2> $O/bench_gain$g.err; done
python bench.py --config stress --whole-frame --steps 60 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_stress_whole.json 2> $O/bench_stress_whole.err
ls -la $O
