#!/usr/bin/env python3
"""Train-step throughput (BASELINE config 4 shape: synthetic DNS-style mixes, 4 s clips, Adam, clip 3.0).

    python tools/train_bench.py --batch 512 --steps 5 --warmup 2 [--profile]

fp32 end to end (the reference trains in fp32; config 4 asks for bf16, which this path does not use).
Prints one JSON line: frames/s of full train steps (STFT x2, forward, HybridLoss, backward, clip, Adam)
and the split model-forward / model-backward / rest measured with HIP events."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    import numpy as np
    import torch
    import __graft_entry__ as graft
    graft.build()
    import gtcrn_micro_amd as G
    from gtcrn_micro_amd.train import make_training, synthetic_mix, train_step
    torch.manual_seed(43)
    model, opt, sched, loss_func = make_training(device="cuda")
    model.train()
    L = int(a.seconds * 16000)
    noisy, clean = synthetic_mix(a.batch, samples=L, seed=43)
    T = 1 + L // 256
    ws = G.Trainer.workspace_bytes(a.batch, T)
    for _ in range(a.warmup):
        train_step(model, opt, sched, loss_func, noisy, clean)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, gn = train_step(model, opt, sched, loss_func, noisy, clean)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / a.steps
    # split: model forward / backward alone
    hann = torch.hann_window(512, device="cuda")
    spec = G.stft(noisy, hann)
    tr = model._trainer(spec.device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    gout = torch.randn_like(spec) * 0.01
    ev[0].record(); out = tr.forward(model._flat, spec); ev[1].record()
    tr.backward(model._flat, spec, gout); ev[2].record()
    torch.cuda.synchronize()
    print(json.dumps({
        "metric": "train frames/sec (STFT x2 -> forward -> HybridLoss -> backward -> clip -> Adam)",
        "value": round(a.batch * T / el, 1), "unit": "frames/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(el * 1e3, 3), "dtype": "f32", "data": "synthetic DNS-style mixes",
        "config": {"workload": f"train step, B={a.batch} clips x {a.seconds:g} s (T={T}), fp32, Adam, clip 3.0"},
        "model_forward_ms": round(ev[0].elapsed_time(ev[1]), 3), "model_backward_ms": round(ev[1].elapsed_time(ev[2]), 3),
        "workspace_GB": round(ws / 2 ** 30, 2), "loss": float(loss), "grad_norm": float(gn)}))


if __name__ == "__main__":
    main()
