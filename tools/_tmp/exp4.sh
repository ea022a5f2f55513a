for d in 0 1 2; do
  echo "== PN_NERF_DBG=$d"
  PN_NERF_DBG=$d timeout 300 python tools/time_net.py 2>&1 | tail -1
done
