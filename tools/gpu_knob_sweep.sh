#!/bin/bash
# kernel stats of the eager chair frame for a few settings of an environment knob: tools/gpu_knob_sweep.sh VAR v1 v2 ...
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r03c; mkdir -p $O
VAR=$1; shift
for v in "$@"; do
  cd /tmp; rm -rf /tmp/ks
  env $VAR=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o eager --output-format csv -- python $R/tools/run_frames.py --frames 20 --no-counters --no-sim > /tmp/ks.log 2>&1
  find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/stats_${VAR}_$v.csv \;
  cd $R
  echo "== $VAR=$v"; python - <<PY
import csv
rows=list(csv.DictReader(open('$O/stats_${VAR}_$v.csv')))
tot=0
for r in rows:
    n=r['Name']
    if n.startswith('k_frame') or 'k_comp' in n or 'k_march' in n or 'k_nerf' in n or 'k_list' in n or 'k_get_rays' in n:
        tot+=float(r['TotalDurationNs'])/20e3
        print(f"{n[:36]:36s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f} min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f} /frame {float(r['TotalDurationNs'])/20e3:7.1f}")
print('render kernels per frame us', round(tot,1))
PY
done
