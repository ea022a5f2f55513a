#!/bin/bash
# usage: gpu_variants.sh <outdir> <variant...>   ("base" = the shipped library); per variant: bench line + eager per-kernel times
export TMPDIR=/tmp
O=gpurun_out/$1; shift
mkdir -p $O
for v in "$@"; do
  if [ $v = base ]; then unset PN_LIB_PATH; else export PN_LIB_PATH=$PWD/pienerf_amd/lib/variants/$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 150 > $O/bench_$v.json 2> $O/e_$v
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_$v.json'))
    print('$v', d['value'], d['ms_per_step'], 'march', d['roofline']['ms_per_frame'], d['breakdown_ms']['march_per_trip'], 'eager', d['breakdown_ms']['render_frame_eager'])
except Exception as e:
    print('$v failed', e, open('$O/e_$v').read()[-300:])
PY
  rm -rf /tmp/prof; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o eager -- python $OLDPWD/tools/run_frames.py --frames 3 > /dev/null 2>&1)
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/prof/**/*kernel_trace.csv', recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    d=collections.defaultdict(list)
    for r in rows:
        d[r['Kernel_Name'][:30]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
    for n,v in d.items():
        if any(k in n for k in ('march',)): print('   ', n, [round(x,1) for x in (v[-16:-11] if len(v) > 16 else v[-2:-1])])
PY
done
