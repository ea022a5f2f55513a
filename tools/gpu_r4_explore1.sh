#!/bin/bash
# round 4, first look: how the pipelined chair bench scales with render lanes, whether the 4-lane collapse is the polling composite or the
# hardware queues, and the per-lane kernel timeline of the 3-lane run
export TMPDIR=/tmp
OUT=gpurun_out/r4x1
mkdir -p $OUT
B="python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20"
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['value_unprimed'], d['config']['frames_continued_past_captured_trips'])"; }
for L in 1 2 3; do echo "lanes=$L $($B --lanes $L 2>$OUT/err_l$L.txt | val)"; done | tee $OUT/lanes.txt
echo "lanes=4 $($B --lanes 4 2>$OUT/err_l4.txt | val)" | tee -a $OUT/lanes.txt
echo "lanes=4 split_compact $(PN_SPLIT_COMPACT=1 $B --lanes 4 2>$OUT/err_l4s.txt | val)" | tee -a $OUT/lanes.txt
echo "lanes=3 split_compact $(PN_SPLIT_COMPACT=1 $B --lanes 3 2>$OUT/err_l3s.txt | val)" | tee -a $OUT/lanes.txt
echo "lanes=4 depth=1 $($B --lanes 4 --depth 1 2>$OUT/err_l4d1.txt | val)" | tee -a $OUT/lanes.txt
echo "lanes=4 depth=1 split $(PN_SPLIT_COMPACT=1 $B --lanes 4 --depth 1 2>$OUT/err_l4d1s.txt | val)" | tee -a $OUT/lanes.txt
echo "lanes=6 depth=1 split $(PN_SPLIT_COMPACT=1 $B --lanes 6 --depth 1 2>$OUT/err_l6d1s.txt | val)" | tee -a $OUT/lanes.txt
echo "lanes=3 no-d2h $($B --lanes 3 --no-d2h 2>/dev/null | val)" | tee -a $OUT/lanes.txt
# kernel trace of the 3-lane pipeline
rm -rf /tmp/prof_l3; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_l3 -o t -- python $OLDPWD/bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 10 --prime 10 --lanes 3 > $OLDPWD/$OUT/bench_l3_prof.json 2>/dev/null)
python tools/lane_timeline.py /tmp/prof_l3 0.3 -v > $OUT/lane_timeline_l3.txt 2>&1
python tools/pipe_trace.py /tmp/prof_l3 0.3 > $OUT/pipe_trace_l3.txt 2>&1
rm -rf /tmp/prof_l4; (cd /tmp && PN_SPLIT_COMPACT=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_l4 -o t -- python $OLDPWD/bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 10 --prime 10 --lanes 4 > $OLDPWD/$OUT/bench_l4_prof.json 2>/dev/null)
python tools/lane_timeline.py /tmp/prof_l4 0.3 > $OUT/lane_timeline_l4.txt 2>&1
python tools/pipe_trace.py /tmp/prof_l4 0.3 > $OUT/pipe_trace_l4.txt 2>&1
tail -3 $OUT/lane_timeline_l3.txt
