for cfg in "" "PN_GATHER_SLOTS=8" "PN_GATHER_SLOTS=16" "PN_ELASTIC_WAVES=6" "PN_ELASTIC_WAVES=7" "PN_GATHER_SLOTS=8 PN_ELASTIC_WAVES=6" "PN_GATHER_SLOTS=16 PN_ELASTIC_WAVES=6"; do
  echo "== $cfg"
  env $cfg timeout 200 python tools/pipe_probe.py --lanes 3 --steps 400 2>&1 | grep lanes=
done
