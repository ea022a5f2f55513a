#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4halfbwd; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_trainloop.py tests/test_gpu_half.py -m gpu -q -x 2>&1 | tail -30 | tee $OUT/pytest.txt
