for mlp in 2 1; do
  echo "== MLP=$mlp (1 = bf16 split, 2 = f32)"
  PN_NERF_MLP=$mlp timeout 300 python tools/time_net.py 2>&1 | tail -1 | cut -c1-70
  PN_NERF_MLP=$mlp PN_NERF_DBG=1 timeout 300 python tools/time_net.py 2>&1 | tail -1 | cut -c1-70
  PN_NERF_MLP=$mlp timeout 200 python tools/pipe_probe.py --lanes 1 3 --steps 400 2>&1 | grep lanes=
done
PN_NERF_MLP=1 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12
