#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4forms; mkdir -p $OUT
B="python bench.py --no-extras --no-cpu-baseline"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['breakdown_ms']; print(d['value'], d['value_unprimed'], d['verified'], d['config']['frames_continued_past_captured_trips'])"; }
for L in 2 3; do for F in whole fold plain; do echo "lanes=$L form=$F $($B --lanes $L --form $F 2>/dev/null | val)" | tee -a $OUT/forms.txt; done; done
for G in 85 96 112; do echo "lanes=3 form=whole grid=$G $(PN_FUSED_GRID=$G $B --lanes 3 --form whole 2>/dev/null | val)" | tee -a $OUT/forms.txt; done
for G in 160 192; do echo "lanes=2 form=whole grid=$G $(PN_FUSED_GRID=$G $B --lanes 2 --form whole 2>/dev/null | val)" | tee -a $OUT/forms.txt; done
echo "lanes=2 form=whole K20 $($B --lanes 2 --form whole --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/forms.txt
echo "lanes=3 form=auto K20 $($B --steps 20 --warmup 5 2>/dev/null | val)" | tee -a $OUT/forms.txt
