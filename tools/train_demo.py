#!/usr/bin/env python3
"""Short training run on synthetic DNS-style mixes (the reference's train step, train.py:239-288, on the HIP path):
random initial weights, Adam, warm-up-cosine schedule, clip 3.0.  Prints the loss curve as JSON.

    python tools/train_demo.py --steps 300 --batch 64 [--storage bf16]

--storage bf16 keeps the saved activations in bf16 (BASELINE configs[3]); run both and compare the curves.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--storage", choices=["f32", "bf16", "bf16_saves", "bf16_grads"], default="f32")
    a = ap.parse_args()
    import torch
    from gtcrn_micro_amd.train import make_training, synthetic_mix, train_step
    torch.manual_seed(43)
    model, opt, sched, loss_func = make_training({"warmup_steps": 50, "decay_until_step": a.steps, "max_lr": 2e-3},
                                                 device="cuda")
    model.train()
    model.set_activation_storage(a.storage)
    sets = [synthetic_mix(a.batch, samples=int(a.seconds * 16000), seed=100 + i) for i in range(8)]
    curve = []
    t0 = time.perf_counter()
    for step in range(a.steps):
        noisy, clean = sets[step % len(sets)]
        loss, gn = train_step(model, opt, sched, loss_func, noisy, clean)
        if step % 10 == 0 or step == a.steps - 1:
            curve.append({"step": step, "loss": round(float(loss), 4), "grad_norm": round(float(gn), 3),
                          "lr": opt.param_groups[0]["lr"]})
    torch.cuda.synchronize()
    # eval-mode check on held-out mixes with the trained weights (running statistics) through the inference kernels
    model.eval()
    noisy, clean = synthetic_mix(a.batch, samples=int(a.seconds * 16000), seed=999)
    import gtcrn_micro_amd as G
    win = torch.hann_window(512, device="cuda")
    with torch.no_grad():
        ev_before = float(loss_func(G.stft(noisy, win), G.stft(clean, win)))
        ev_after = float(loss_func(model(G.stft(noisy, win)), G.stft(clean, win)))
    print(json.dumps({"storage": a.storage, "steps": a.steps, "batch": a.batch, "seconds_per_clip": a.seconds,
                      "wall_s": round(time.perf_counter() - t0, 2), "curve": curve,
                      "heldout_loss_noisy_input": round(ev_before, 4), "heldout_loss_enhanced": round(ev_after, 4)}))


if __name__ == "__main__":
    main()
