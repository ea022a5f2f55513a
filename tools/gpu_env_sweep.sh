#!/bin/bash
# usage: gpu_env_sweep.sh VAR v1 v2 ...   eager per-kernel march times + per-trip records for each value of an environment variable
export TMPDIR=/tmp
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  rm -rf /tmp/prof; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o eager -- python $OLDPWD/tools/run_frames.py --frames 3 2>/dev/null | grep trips | cut -c1-400)
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/prof/**/*kernel_trace.csv', recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    d=collections.defaultdict(list)
    for r in rows:
        d[r['Kernel_Name'][:30]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
    for n,v in d.items():
        if 'march' in n: print('  $VAR=$v', n, [round(x,1) for x in (v[-16:-11] if len(v) > 16 else v[-2:-1])], round(sum(v[-16:-11]) if len(v)>16 else v[-2],1))
PY
done
