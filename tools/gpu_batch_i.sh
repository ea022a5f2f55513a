#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02v
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_half.py -m gpu -q -x 2>&1 | tail -5
python bench.py --no-cpu-baseline --no-extras --steps 150 > $O/bench.json 2> $O/e
python - <<PY
import json
try:
    d=json.load(open('$O/bench.json'))
    print('bench', d['value'], d['ms_per_step'], 'march', d['roofline']['ms_per_frame'], d['breakdown_ms']['march_per_trip'], 'eager', d['breakdown_ms']['render_frame_eager'])
except Exception as e:
    print('failed', e, open('$O/e').read()[-300:])
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o eager -- python $OLDPWD/tools/run_frames.py --frames 3 > $OLDPWD/$O/time_frame.txt 2>&1; cd $OLDPWD
python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/prof/**/*kernel_trace.csv', recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    d=collections.defaultdict(list)
    for r in rows:
        d[r['Kernel_Name'][:44]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
    for n,v in d.items():
        if any(k in n for k in ('march','nerf','frame','composite','compact')): print(n, len(v), [round(x,1) for x in v[-10:]])
PY
