#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r4rccl
timeout 900 python -m pytest tests/test_frames_gpu.py -x -q -k "rccl or single_rank" 2>&1 | tail -25 | tee gpurun_out/r4rccl/pytest.txt
