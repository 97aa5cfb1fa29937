#!/usr/bin/env python3
"""A/B timing of a kernel experiment on ONE GPU box: the default library against the `-DGT_EXP` build.

    python tools/ab_bench.py [--rounds 3] [--batch 256]

Box-to-box spread on this pool is ~2 %, more than most single optimisations are worth, so an experiment is put
behind `#ifdef GT_EXP` in the sources, both libraries are built here (gtcrn_micro_amd.build, exp=True) and the
bench workload is run alternately through each in child processes (GTCRN_LIB_VARIANT=exp selects the second).
Prints per-kernel milliseconds per round and the means.  Diagnostic only; nothing in the product path reads GT_EXP.
Standing switches for the exp build (GT_EXP_FLAGS): -DGT_F32_DENSE = the round-2 fp32-MFMA dense 3x3 and de_convs.3 of
the decoder instead of the bf16 split forms (profiles/r03_ab_r3_vs_r2_matrix_path.txt).
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(variant, batch):
    env = dict(os.environ)
    env.pop("GTCRN_LIB_VARIANT", None)
    if variant:
        env["GTCRN_LIB_VARIANT"] = variant
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-secondary", "--no-cpu-baseline", "--batch", str(batch)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, check=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    k = dict(d["kernel_ms"])
    k["step"] = d["ms_per_step"]
    return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    from gtcrn_micro_amd.build import build_native
    build_native()
    build_native(exp=True)
    acc = {"base": [], "exp": []}
    for r in range(a.rounds):
        for name, variant in (("base", None), ("exp", "exp")):
            k = run(variant, a.batch)
            acc[name].append(k)
            print(f"round {r} {name:4s} " + "  ".join(f"{n}={v:.4f}" for n, v in k.items()), flush=True)
    for name in ("base", "exp"):
        keys = acc[name][0].keys()
        print(f"mean    {name:4s} " + "  ".join(f"{n}={sum(x[n] for x in acc[name]) / len(acc[name]):.4f}" for n in keys))


if __name__ == "__main__":
    main()
