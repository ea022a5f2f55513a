for mlp in 0 1; do for d in 0 1 2; do
  echo "== MLP=$mlp DBG=$d"
  PN_NERF_MLP=$mlp PN_NERF_DBG=$d timeout 300 python tools/time_net.py 2>&1 | tail -1 | cut -c1-70
done; done
