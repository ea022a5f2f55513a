for a in "20 5" "20 0" "200 20"; do set -- $a; python bench.py --no-cpu-baseline --no-extras --steps $1 --warmup $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps $1 warmup $2', d['value'], round(d['ms_per_step']*d['steps'],2),'ms total')"; done
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 --prime 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('prime 200 steps 20', d['value'])"
