#!/bin/bash
export TMPDIR=/tmp
R=$PWD
cd /tmp
AMD_LOG_LEVEL=4 python $R/tools/d2h_probe.py 2> /tmp/amdlog.txt | tail -1
grep -i "sdma\|HSA Copy\|blit\|engine" /tmp/amdlog.txt | sort | uniq -c | sort -rn | head -20
for v in 1 2 3; do GPU_BLIT_ENGINE_TYPE=$v python $R/tools/d2h_probe.py 2>&1 | tail -1; done
HSA_ENABLE_SDMA=1 HSA_FORCE_SDMA_SIZE=1 python $R/tools/d2h_probe.py 2>&1 | tail -1
rocminfo | grep -i -A3 "sdma\|engine" | head -20
