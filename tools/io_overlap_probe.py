#!/usr/bin/env python3
"""Which pairs of {copy-in, kernels, copy-out} actually overlap on this box?  (bench.py's `io` object reports the served
rate; this probe says where a shortfall against the link rate comes from.)  Prints one JSON object.
    python tools/io_overlap_probe.py            # default runtime settings
    HSA_ENABLE_SDMA=0 python tools/io_overlap_probe.py   # copies as blit kernels instead of SDMA engines"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gtcrn_micro_amd import Engine

params = np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32)
eng = Engine(params, 0)
B, L = 256, 64000
T = 1 + L // 256
eng.reserve(B, T)
win = torch.hann_window(512).pow(0.5).cuda()
x = torch.randn(B, L, device="cuda") * 0.1
y = torch.empty(B, 256 * (T - 1), device="cuda")
x2, y2 = torch.empty_like(x), torch.empty_like(y)
hin = torch.empty(x.shape, pin_memory=True)
hout = torch.empty(y.shape, pin_memory=True)
s_in, s_cmp, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
nchunk = int(os.environ.get("CHUNKS", "1"))


def cin():
    with torch.cuda.stream(s_in):
        for c in range(nchunk):
            lo, hi = c * B // nchunk, (c + 1) * B // nchunk
            x2[lo:hi].copy_(hin[lo:hi], non_blocking=True)


def cout():
    with torch.cuda.stream(s_out):
        for c in range(nchunk):
            lo, hi = c * B // nchunk, (c + 1) * B // nchunk
            hout[lo:hi].copy_(y2[lo:hi], non_blocking=True)


def comp():
    with torch.cuda.stream(s_cmp):
        eng.forward_wave(x, win, out=y)


def ms(fns, n=20):
    for f in fns:
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in fns:
            f()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 4)


for _ in range(30):
    comp()
torch.cuda.synchronize()
res = {"HSA_ENABLE_SDMA": os.environ.get("HSA_ENABLE_SDMA"), "chunks": nchunk,
       "in": ms([cin]), "out": ms([cout]), "kernels": ms([comp]),
       "in+out": ms([cin, cout]), "in+kernels": ms([cin, comp]), "kernels+out": ms([comp, cout]),
       "in+kernels+out": ms([cin, comp, cout])}
res["sum_serial"] = round(res["in"] + res["out"] + res["kernels"], 4)
print(json.dumps(res))
