#!/bin/bash
# usage: frame_trace.sh [run_frames.py args]   kernel-by-kernel timeline of the last eager frame (start offset, duration, gap to the previous kernel, us)
export TMPDIR=/tmp
rm -rf /tmp/prof_ft; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ft -o t -- python $OLDPWD/tools/run_frames.py --frames 4 --no-counters "$@" 2>/dev/null | tail -2 | cut -c1-300)
python - <<PY
import csv,glob
f=glob.glob('/tmp/prof_ft/**/*kernel_trace.csv', recursive=True)
rows=sorted(csv.DictReader(open(f[0])), key=lambda r:int(r['Start_Timestamp']))
# the last frame starts at the last k_frame_tables
idx=[i for i,r in enumerate(rows) if 'k_frame_tables' in r['Kernel_Name']]
i0=idx[-1]
# include the simulator kernels enqueued around it: take everything from the last k_step_begin before i0, if close
t0=int(rows[i0]['Start_Timestamp']); prev_end=t0
print('   start     dur     gap  kernel')
tot=0
for r in rows[i0:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:8.1f} {(e-s)/1e3:7.1f} {(s-prev_end)/1e3:7.1f}  {r['Kernel_Name'][:70]}  grid={r.get('Grid_Size_X','?')} wg={r.get('Workgroup_Size_X','?')} lds={r.get('LDS_Block_Size','?')} vgpr={r.get('VGPR_Count','?')}")
    prev_end=max(prev_end,e); tot+=e-s
print('span us', (prev_end-t0)/1e3, 'sum of kernels us', tot/1e3)
PY
