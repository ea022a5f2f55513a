#!/usr/bin/env python
"""How the per-frame device-to-host copy (12.8 MB into pinned memory) is executed: time and engine.
    python tools/d2h_probe.py            (prints GB/s of an idle copy; under rocprofv3 --kernel-trace --memory-copy-trace the trace shows the engine)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = 640000 * 5
dev = torch.device("cuda:0")
src = torch.randn(N, device=dev)
dst = torch.empty(N).pin_memory()
s = torch.cuda.Stream()
for name, fn in (("torch copy_ non_blocking", lambda: dst.copy_(src, non_blocking=True)),):
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20):
            fn()
        e1.record(s)
        s.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: {ms * 1e3:.1f} us per 12.8 MB copy = {N * 4 / ms / 1e6:.1f} GB/s", {k: os.environ.get(k) for k in ("HSA_ENABLE_SDMA", "GPU_FORCE_BLIT_COPY_SIZE", "GPU_BLIT_ENGINE_TYPE")})
assert torch.equal(dst, src.cpu())
