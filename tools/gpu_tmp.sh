for c in stress trex; do
  python bench.py --no-cpu-baseline --no-extras --config $c 2>/tmp/e_$c | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'], d['roofline'].get('ms_per_frame'), d['breakdown_ms'].get('march_per_trip'), d['config'].get('launch','')[:60])" || tail -3 /tmp/e_$c
done
python bench.py --no-cpu-baseline --no-extras --config stress --whole-frame 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('stress whole-frame', d['value'], d['ms_per_step'], d['roofline'].get('ms_per_frame'))"
