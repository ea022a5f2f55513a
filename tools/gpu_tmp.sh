for a in "--lanes 3 --depth 1" "--lanes 3 --depth 2" "--lanes 3 --depth 1" "--lanes 3 --depth 2" "--lanes 3 --depth 1 --no-d2h" "--lanes 2 --depth 1 --copy-on copy" "--lanes 3 --depth 1 --steps 20 --warmup 5"; do python bench.py --no-cpu-baseline --no-extras $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a', d['value'])"; done
