for i in 1 2 3; do echo "== run $i"; PN_NERF_MLP=2 timeout 300 python tools/_tmp/repro.py 2>&1 | grep -v amdgpu.ids; done
echo "== bf16"; PN_NERF_MLP=1 timeout 300 python tools/_tmp/repro.py 2>&1 | grep -v amdgpu.ids
