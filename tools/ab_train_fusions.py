#!/usr/bin/env python3
"""Same-box A/B of the train step's pass fusions: full steps (forward, HybridLoss, backward, clip, Adam) at B = 512 x 4 s
for each fusion mask given, interleaved, a few rounds.

    python tools/ab_train_fusions.py [--masks 7,15,31,63,127,255,511,1023,2047,4095,8191,16383,32767,65535] [--storage f32] [--steps 6] [--rounds 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--masks", default="7,15,31,63,127,255,511,1023,2047,4095,8191,16383,32767,65535")
    ap.add_argument("--storage", default="f32")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    import torch
    from gtcrn_micro_amd.train import make_training, synthetic_mix, train_step
    masks = [int(m) for m in a.masks.split(",")]
    torch.manual_seed(43)
    model, opt, sched, loss_func = make_training(device="cuda")
    model.train()
    if a.storage != "f32":
        model.set_activation_storage(a.storage)
    noisy, clean = synthetic_mix(a.batch, samples=64000, seed=43)
    tr = model._trainer(noisy.device)
    res = {m: [] for m in masks}
    for rnd in range(a.rounds + 1):
        for m in masks:
            tr.set_fusions(m)
            train_step(model, opt, sched, loss_func, noisy, clean)          # re-plan + warm
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                train_step(model, opt, sched, loss_func, noisy, clean)
            torch.cuda.synchronize()
            if rnd:
                res[m].append((time.perf_counter() - t0) / a.steps * 1e3)
    for m in masks:
        v = sorted(res[m])
        print(f"fusions {m:>2} [{a.storage}]: {v[len(v) // 2]:.3f} ms per step (min {v[0]:.3f}, max {v[-1]:.3f})")


if __name__ == "__main__":
    main()
