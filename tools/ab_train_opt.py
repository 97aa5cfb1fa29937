#!/usr/bin/env python3
"""Same-box A/B of the train step's optimizer: utils.optim.FlatAdam (clip + Adam in one launch) against
torch.nn.utils.clip_grad_norm_ + torch.optim.Adam, interleaved, B = 512 x 4 s.  One JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gtcrn_micro_amd.train import make_training, synthetic_mix, train_step

B = int(os.environ.get("B", "512"))
noisy, clean = synthetic_mix(B, samples=64000, seed=43)
ctx = {}
for fused in (True, False):
    torch.manual_seed(43)
    m, o, s, l = make_training(device="cuda", fused_optimizer=fused)
    m.train()
    if os.environ.get("STORAGE"):
        m.set_activation_storage(os.environ["STORAGE"])
    ctx[fused] = (m, o, s, l)
    for _ in range(2):
        train_step(m, o, s, l, noisy, clean)
torch.cuda.synchronize()
res = {True: [], False: []}
for rep in range(4):
    for fused in (True, False):
        m, o, s, l = ctx[fused]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            train_step(m, o, s, l, noisy, clean)
        torch.cuda.synchronize()
        res[fused].append((time.perf_counter() - t0) / 5 * 1e3)
print(json.dumps({"B": B, "storage": os.environ.get("STORAGE", "f32"),
                  "flat_adam_ms": [round(x, 3) for x in res[True]], "torch_adam_ms": [round(x, 3) for x in res[False]],
                  "flat_adam_min": round(min(res[True]), 3), "torch_adam_min": round(min(res[False]), 3)}))
