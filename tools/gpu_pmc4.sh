#!/bin/bash
export TMPDIR=/tmp
PAT=$1; shift
R=$PWD
cd /tmp
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_$i -o p --output-format csv -- python $R/tools/run_frames.py --frames 2 --no-sim > /tmp/pmc_$i.log 2>&1 || { echo "pass $set failed/timeout"; tail -2 /tmp/pmc_$i.log; continue; }
  python $R/tools/pmc_dispatch.py /tmp/pmc_$i "$PAT" 8 5
done
