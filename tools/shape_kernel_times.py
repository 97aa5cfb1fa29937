#!/usr/bin/env python3
"""Per-kernel HIP-event times of the offline wave -> wave path at arbitrary batch shapes (GPU box).

    python tools/shape_kernel_times.py 1x31 1x4 32x4 256x4 257x4 512x4      # B x seconds
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gtcrn_micro_amd import Engine

params = np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32)
eng = Engine(params, 0)
win = torch.hann_window(512).pow(0.5).cuda()
for shape in sys.argv[1:] or ["256x4", "257x4"]:
    B, sec = shape.split("x")
    B, L = int(B), int(float(sec) * 16000)
    T = 1 + L // 256
    x = torch.randn(B, L, device="cuda") * 0.1
    y = torch.empty(B, 256 * (T - 1), device="cuda")
    eng.reserve(B, T)
    for _ in range(10):
        eng.forward_wave(x, win, out=y)
    eng.timing_enable(True)
    for _ in range(30):
        eng.forward_wave(x, win, out=y)
    torch.cuda.synchronize()
    k = eng.timing_read()
    eng.timing_enable(False)
    print(shape, json.dumps({n: round(v[0], 4) for n, v in k.items()}), "sum", round(sum(v[0] for v in k.values()), 4))
