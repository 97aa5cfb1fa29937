#!/usr/bin/env python3
"""Bit-identity check of a kernel experiment: the default library against the `-DGT_EXP` build on the same inputs.

    python tools/ab_check.py            # builds both, runs each in a child process, compares output digests

Companion of tools/ab_bench.py (which times the two): an experiment that reorders work must not change a bit of the
offline forward.  Prints one sha256 per shape and variant; exit code 1 on a mismatch.
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(256, 64000), (256, 40000), (64, 64000), (300, 48000), (512, 16000)]


def child():
    import numpy as np
    import torch
    import gtcrn_micro_amd as G
    params = np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32)
    eng = G.Engine(params, 0)
    win = torch.hann_window(512).pow(0.5).cuda()
    for B, L in SHAPES:
        rng = np.random.default_rng(B + L)
        x = torch.from_numpy((rng.standard_normal((B, L)) * 0.1).astype(np.float32)).cuda()
        y = eng.forward_wave(x, win)
        torch.cuda.synchronize()
        print(f"DIGEST {B}x{L} {hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()}", flush=True)


def main():
    if os.environ.get("GT_AB_CHILD"):
        return child()
    from gtcrn_micro_amd.build import build_native
    build_native()
    build_native(exp=True)
    out = {}
    for variant in (None, "exp"):
        env = dict(os.environ, GT_AB_CHILD="1")
        env.pop("GTCRN_LIB_VARIANT", None)
        if variant:
            env["GTCRN_LIB_VARIANT"] = variant
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, check=True)
        out[variant] = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")]
        for ln in out[variant]:
            print(variant or "base", ln)
    same = out[None] == out["exp"] and len(out[None]) == len(SHAPES)
    print("bit-identical" if same else "MISMATCH")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
