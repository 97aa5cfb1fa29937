#!/usr/bin/env python3
"""Diagnostic (GPU box): per-kernel HIP-event times of one-frame streaming steps for 1024 streams and, with
GTCRN_LIB_VARIANT=stamps, the per-phase cycle sums of the diagnostic build.

    python tools/stream_kernel_times.py
"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtcrn_micro_amd import Engine
params = np.fromfile("tests/golden/params_dns3.f32", dtype=np.float32)
eng = Engine(params, 0)
N = 1024
spec = (torch.randn(N, 64, 257, 2, device="cuda") * 0.3).permute(0, 2, 1, 3)
out = torch.empty((N, 64, 257, 2), device="cuda").permute(0, 2, 1, 3)
state = eng.new_state(N)
eng.reserve(N, 1)
for t in range(8):
    eng.stream_step(state, spec[:, :, t:t + 1], out=out[:, :, t:t + 1])
torch.cuda.synchronize()
eng.timing_enable(True)
for t in range(8, 40):
    eng.stream_step(state, spec[:, :, t:t + 1], out=out[:, :, t:t + 1])
torch.cuda.synchronize()
print({k: (round(v[0] * 1e3, 1), v[1]) for k, v in eng.timing_read().items()})
eng.timing_enable(False)
if os.environ.get("GTCRN_LIB_VARIANT") == "stamps":
    eng.debug_enable(True)
    eng.stream_step(state, spec[:, :, 41:42], out=out[:, :, 41:42])
    torch.cuda.synchronize()
    for k in (0, 1, 3):
        st = eng.stamps(k, N).astype(np.float64).mean(axis=0)
        print(k, [int(x) for x in st])
