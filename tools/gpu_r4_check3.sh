#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4check2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_gpu2.txt
