O=gpurun_out/r02final; mkdir -p $O
python bench.py > $O/bench_chair.json 2> $O/bench_chair.err
python bench.py --config stress > $O/bench_stress.json 2> $O/bench_stress.err
python bench.py --config trex > $O/bench_trex.json 2> $O/bench_trex.err
python -c "
import json
for c in ('chair','stress','trex'):
    d=json.load(open('$O/bench_%s.json'%c)); print(c, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'])
"
