for mlp in 0 1; do
  echo "== PN_NERF_MLP=$mlp"
  PN_NERF_MLP=$mlp timeout 300 python tools/time_net.py 2>&1 | tail -1
done
PN_NERF_MLP=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for mlp in 0 1; do
  PN_NERF_MLP=$mlp timeout 200 python tools/pipe_probe.py --lanes 1 3 --steps 400 2>&1 | grep lanes=
done
