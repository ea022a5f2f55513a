for v in base cap1250; do
  if [ $v = base ]; then unset PN_LIB_PATH; else export PN_LIB_PATH=$PWD/pienerf_amd/lib/variants/$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['ms_per_frame'])"
  python bench.py --no-cpu-baseline --no-extras --steps 300 --no-d2h 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v no-d2h', d['value'])"
done
