#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py 2>&1 | tail -150 > $O/pytest_gpu.txt
tail -100 $O/pytest_gpu.txt
