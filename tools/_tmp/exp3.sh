for x in 0 1; do
  echo "== PN_GRID_XCD=$x"
  PN_GRID_XCD=$x PN_NERF_MLP=0 timeout 300 python tools/time_net.py 2>&1 | tail -1
done
PN_NERF_MLP=1 timeout 900 python -m pytest tests/test_frames_gpu.py -m gpu -x -q 2>&1 | tail -30
