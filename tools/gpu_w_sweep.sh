#!/bin/bash
export TMPDIR=/tmp
for W in "$@"; do
  rm -rf /tmp/prof; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o eager -- python $OLDPWD/tools/run_frames.py --frames 3 --W $W 2>/dev/null | grep trips | cut -c1-300)
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/prof/**/*kernel_trace.csv', recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    d=collections.defaultdict(list)
    for r in rows:
        d[r['Kernel_Name'][:30]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
    for n,v in d.items():
        if 'march' in n or 'nerf' in n: print('  W=$W', n, [round(x,1) for x in (v[-16:-11] if len(v) > 16 else v[-2:-1])])
PY
done
