#!/bin/bash
# A/B of pn_render_opts.ray_tile_w (8 x 8 pixel tile order of the alive list) against the row-major order (PN_RAY_TILE_OFF=1) and against
# 16 x 4 / 32 x 2 tiles (PN_RAY_TILE_LOG2W=4 / 5), alternating runs.  Output: gpurun_out/r03tile3/
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r03tile3; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-extras > $O/bench_chair_$name.json 2>/dev/null
  env "$@" python bench.py --no-cpu-baseline --no-extras --lanes 1 > $O/bench_lanes1_$name.json 2>/dev/null
}
for rep in 1 2; do
  run off_$rep PN_RAY_TILE_OFF=1
  run t8x8_$rep PN_RAY_TILE_LOG2W=3
  run t16x4_$rep PN_RAY_TILE_LOG2W=4
  run t32x2_$rep PN_RAY_TILE_LOG2W=5
done
for cfg in trex stress; do
  PN_RAY_TILE_OFF=1 python bench.py --no-cpu-baseline --no-extras --config $cfg --whole-frame > $O/bench_${cfg}_off.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-extras --config $cfg --whole-frame > $O/bench_${cfg}_t8x8.json 2>/dev/null
done
st() { name=$1; shift
  cd /tmp; rm -rf /tmp/st_$name
  env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o s -- python $R/tools/run_frames.py --frames 20 --no-counters > /dev/null 2>&1
  find /tmp/st_$name -name "*kernel_stats.csv" -exec cp {} $O/eager_${name}_kernel_stats.csv \;
  cd $R
}
st off PN_RAY_TILE_OFF=1; st t8x8 PN_RAY_TILE_LOG2W=3; st t16x4 PN_RAY_TILE_LOG2W=4; st off2 PN_RAY_TILE_OFF=1; st t8x8b PN_RAY_TILE_LOG2W=3
python - <<PY
import json,glob,os,csv
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.load(open(f)); b=d.get('breakdown_ms',{})
        print(os.path.basename(f), d['value'], d.get('value_unprimed'), d['ms_per_step'], 'march', d['roofline'].get('ms_per_frame'), 'net', d.get('network',{}).get('ms_per_frame'), b.get('network_per_trip'), b.get('march_per_trip'))
    except Exception as e: print(f,'ERR',e)
for f in sorted(glob.glob('$O/eager_*_kernel_stats.csv')):
    rows=list(csv.DictReader(open(f))); print(os.path.basename(f), ' '.join(f"{r['Name'][5:22]}={float(r['TotalDurationNs'])/20e3:.1f}" for r in rows if any(k in r['Name'] for k in ('k_march','k_nerf_forward<'))))
PY
