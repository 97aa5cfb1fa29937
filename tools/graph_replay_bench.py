#!/usr/bin/env python3
"""Diagnostic (GPU box): one wave->wave step (B=256 x 4 s) as six plain launches vs the replay of a captured HIP graph."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gtcrn_micro_amd import Engine
params = np.fromfile(os.path.join(ROOT, 'tests', 'golden', 'params_dns3.f32'), dtype=np.float32)
eng = Engine(params, 0)
B, L = 256, 64000
win = torch.hann_window(512, device='cuda').pow(0.5)
wave = torch.randn(B, L, device='cuda') * 0.1
out = torch.empty(B, L, device='cuda')
eng.reserve(B, 251)
for _ in range(10): eng.forward_wave(wave, win, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): eng.forward_wave(wave, win, out=out)
torch.cuda.synchronize(); t_plain = (time.perf_counter() - t0) / 100
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    eng.forward_wave(wave, win, out=out); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        eng.forward_wave(wave, win, out=out)
for _ in range(10): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): g.replay()
torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / 100
print('plain %.4f ms  graph %.4f ms' % (t_plain * 1e3, t_graph * 1e3))
