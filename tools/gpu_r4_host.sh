#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r4host
python tools/host_bound_probe.py --profile 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4host/probe.txt
