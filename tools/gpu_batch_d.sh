#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_frames_gpu.py tests/test_gpu_half.py -m gpu -q -k "pipelined or staged or frame_parallel or harness" 2>&1 | tail -15
python bench.py --no-cpu-baseline > $O/bench_copy.json 2> $O/e1; python bench.py --no-cpu-baseline --no-extras --copy-on lane > $O/bench_lane.json 2> $O/e2
python bench.py --no-cpu-baseline --no-extras --lanes 2 > $O/bench_l2.json 2> $O/e3; python bench.py --no-cpu-baseline --no-extras --lanes 2 --copy-on lane > $O/bench_l2_lane.json 2> $O/e4
python bench.py --no-cpu-baseline --no-extras --lanes 4 --depth 1 --copy-on lane > $O/bench_l4_lane.json 2> $O/e5
python bench.py --config stress --steps 30 --warmup 4 --no-cpu-baseline --no-extras > $O/bench_stress3.json 2> $O/e6
python bench.py --config stress --steps 30 --warmup 4 --no-cpu-baseline --no-extras --staged-streams 4 > $O/bench_stress4.json 2> $O/e7
python bench.py --sigma-gain 0.3 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_gain0.3.json 2> $O/e8
tail -2 $O/e*
