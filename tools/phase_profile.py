#!/usr/bin/env python3
"""Phase anatomy of the model kernels from in-kernel s_memtime stamps (diagnostic build).

    GTCRN_LIB_VARIANT=stamps python tools/phase_profile.py [--batch 256] [--seconds 4]

Builds libgtcrn_micro_hip_stamps.so (-DGT_STAMPS), runs the bench workload once and prints, per
kernel, the average shader cycles a workgroup spends in each barrier-delimited phase.  Read the
SHARES, not the absolute run time: the stamps and their fences perturb the schedule.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# --exp: the stamps inside the `-DGT_EXP` build (the A/B reference of the current kernel experiment; add e.g.
# GT_EXP_FLAGS=-DGT_F32_DENSE for the round-2 fp32-MFMA dense 3x3)
EXP = "--exp" in sys.argv
os.environ["GTCRN_LIB_VARIANT"] = "exp" if EXP else "stamps"
if EXP:
    os.environ["GT_EXP_FLAGS"] = (os.environ.get("GT_EXP_FLAGS", "") + " -DGT_STAMPS").strip()

PHASES = {
    0: {0: "prologue", 10: "A0 stage spec", 11: "A0 fetch next (issue)", 12: "A0 barrier", 13: "A erb bands", 1: "A erb barrier", 2: "B sfe", 3: "C en_conv0", 4: "D en_conv1", 5: "blk pc1", 6: "blk depth+pc2",
        7: "blk tra reduce", 8: "blk apply+store", 9: "chunk end"},
    1: {0: "prologue", 1: "load x", 2: "conv1", 3: "taps+conv3", 4: "ring", 5: "store"},
    3: {0: "prologue", 1: "load x+en4", 5: "blk pc1", 6: "blk dense+pc2", 7: "blk tra reduce", 8: "blk apply+skip",
        9: "de3 stage x (issue)", 10: "de3 stage x (barrier)", 11: "de3+de4 mfma", 15: "Z write (issue)", 12: "Z write (barrier)", 13: "de4 gather+tanh", 14: "bs+mask+store"},
}
PHASES[2] = PHASES[1]
NAMES = ["k_encoder", "k_gtcn1", "k_gtcn2", "k_decoder"]
# --stream N: the single-launch streaming step (k_stream_ms), one row of stamps per workgroup of four streams
STREAM_PHASES = {0: "prologue .. first barrier (params, rings, spec, [mag,re,im])", 1: "ERB bands", 2: "SFE", 3: "en_conv0",
                 4: "en_conv1", 10: "enc blk pc1 (x3)", 11: "enc blk depth+pc2 (x3)", 12: "enc blk TRALite (x3)",
                 9: "GTCN x 2 (8 TCN blocks, per position)", 13: "decoder set-up", 5: "dec blk pc1+split (x3)",
                 6: "dec blk dense+pc2 (x3)", 7: "dec blk TRALite (x3)", 8: "after a block (skip add / permute) (x6)",
                 14: "de_convs.3/4, Z, gather+tanh", 15: "mask + store + epilogue"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--json", default=None)
    ap.add_argument("--exp", action="store_true", help="profile the -DGT_EXP build instead of the default one")
    ap.add_argument("--stream", type=int, default=0, help="profile single-frame streaming steps of this many streams")
    ap.add_argument("--form", type=int, default=2, help="with --stream: 2 = k_stream_ms (four streams per workgroup), 3 = k_stream_wide (seven)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from gtcrn_micro_amd.build import build_native
    build_native(exp=True, force=True) if EXP else build_native(stamps=True)
    from gtcrn_micro_amd import Engine
    params = np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32)
    eng = Engine(params, 0)
    if a.stream:
        N = a.stream
        torch.manual_seed(44)
        spec = (torch.randn(N, 40, 257, 2, device="cuda") * 0.3).permute(0, 2, 1, 3)
        st = eng.new_state(N)
        eng.stream_form(a.form)
        per_wg, kname = (7, "k_stream_wide") if a.form == 3 else (4, "k_stream_ms")
        for t in range(20):
            eng.stream_step(st, spec[:, :, t:t + 1])
        eng.debug_enable(2)                       # stamps only: the step stays ONE launch
        eng.timing_enable(True)
        for t in range(20, 40):
            eng.stream_step(st, spec[:, :, t:t + 1])
        torch.cuda.synchronize()
        kern = eng.timing_read()
        stp = eng.stamps(0, N).astype(np.float64)[: (N + per_wg - 1) // per_wg]   # the LAST step's stamps, one row per workgroup
        avg = stp.mean(axis=0)
        tot = avg.sum()
        us = kern.get(kname, (0, 0))[0] * 1e3
        print(f"\n{kname}, {N} streams: {tot:,.0f} cycles per workgroup (launch {us:.1f} us -> {tot / max(us, 1e-9):.0f} cycles/us)")
        out = {"streams": N, "launch_us": us, "cycles": tot, "phases": {}}
        for i, nm in STREAM_PHASES.items():
            print(f"   {nm:<62} {avg[i]:>10,.0f}  {100 * avg[i] / tot:5.1f} %")
            out["phases"][nm] = avg[i]
        if a.json:
            os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
            json.dump(out, open(a.json, "w"), indent=1)
        return
    B, L = a.batch, int(a.seconds * 16000)
    torch.manual_seed(43)
    wave = torch.randn(B, L, device="cuda") * 0.1
    win = torch.hann_window(512).pow(0.5).cuda()
    for _ in range(3):
        eng.forward_wave(wave, win)
    eng.debug_enable(True)
    eng.timing_enable(True)
    eng.forward_wave(wave, win)
    torch.cuda.synchronize()
    kern = eng.timing_read()
    out = {}
    for k in range(4):
        st = eng.stamps(k, B).astype(np.float64)
        avg = st.mean(axis=0)
        tot = avg.sum()
        if tot == 0:          # offline calls run the barrier-free band GTCN, which carries no stamps
            continue
        if NAMES[k] == "k_encoder" and "k_encoder_gt" in kern:
            kern["k_encoder"] = kern["k_encoder_gt"]   # offline: the front end runs in k_front (no stamps there)
        print(f"\n{NAMES[k]}: {tot:,.0f} cycles per workgroup (launch {kern.get(NAMES[k], (0, 0))[0] * 1e3:.1f} us "
              f"-> {tot / max(kern.get(NAMES[k], (1, 0))[0] * 1e3, 1e-9):.0f} cycles/us)")
        out[NAMES[k]] = {}
        for i, nm in PHASES[k].items():
            print(f"   {nm:<18} {avg[i]:>12,.0f}  {100 * avg[i] / tot:5.1f} %")
            out[NAMES[k]][nm] = avg[i]
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
