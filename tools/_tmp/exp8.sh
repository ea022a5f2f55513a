timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "fused vs oracle|passed|failed|Error" | head
timeout 300 python tools/time_net.py 2>&1 | tail -1 | cut -c1-80
timeout 200 python tools/pipe_probe.py --lanes 1 3 --steps 400 2>&1 | grep lanes=
