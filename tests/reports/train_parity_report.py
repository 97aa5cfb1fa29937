#!/usr/bin/env python3
"""Diagnostic (GPU box): per-stage forward errors and the worst per-tensor gradient errors of the HIP training
path against the reference fixtures, and against the float64 truth (oracle/torch_port.py in double).

    python tests/reports/train_parity_report.py
"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gtcrn_micro_amd as G
import gtcrn_micro_amd._lib as L
GOLD = os.path.join(ROOT, "tests", "golden")
tr = G.Trainer(0)
for tag in ("rand", "dns3"):
    g = np.load(f"{GOLD}/trainstep_{tag}_B3_T12.npz")
    blob = torch.from_numpy(np.fromfile(f"{GOLD}/params_{tag}.f32", np.float32)).cuda()
    spec = torch.from_numpy(g["noisy_spec"]).cuda()
    out = tr.forward(blob, spec)
    torch.cuda.synchronize()
    for name in ("en0", "en1", "en2", "en4", "gtcn1", "gtcn2", "de0", "de2", "de3", "de4"):
        got = tr.tap(name).cpu().numpy(); ref = g["stage:" + name]
        print(tag, name, got.shape, ref.shape, np.abs(got - ref).max() / np.abs(ref).max())
    print(tag, "enh", np.abs(out.cpu().numpy() - g["enh"]).max() / np.abs(g["enh"]).max())
    print(tag, "params_after", np.abs(blob.cpu().numpy() - g["params_after"]).max())
    grads = tr.backward(blob, spec, torch.from_numpy(g["grad_enh"]).cuda()).cpu().numpy()
    ref = g["grads"]
    rows = []
    for name, numel, off in L.param_table():
        r, q = ref[off:off + numel], grads[off:off + numel]
        if not r.any() and not q.any():
            continue
        rows.append((np.abs(q - r).max() / max(np.abs(r).max(), 1e-3 * np.abs(ref).max()), name, np.abs(r).max(), np.abs(q).max()))
    rows.sort(reverse=True)
    for r in rows[:25]:
        print("  %.3e %-50s ref %.3e got %.3e" % r)
    print(tag, "grad norm", np.sqrt((grads.astype(np.float64) ** 2).sum()), np.sqrt((ref.astype(np.float64) ** 2).sum()))
# accuracy against the fp64 truth (same graph in double on the CPU)
from oracle.torch_port import TorchPort
torch.set_num_threads(8)
for tag in ("rand", "dns3"):
    g = np.load(f"{GOLD}/trainstep_{tag}_B3_T12.npz")
    blob_np = np.fromfile(f"{GOLD}/params_{tag}.f32", np.float32)
    _, t64 = TorchPort(blob_np, train=True, dtype=torch.float64).backward_from(g["noisy_spec"], g["grad_enh"])
    blob = torch.from_numpy(blob_np.copy()).cuda()
    spec = torch.from_numpy(g["noisy_spec"]).cuda()
    tr.forward(blob, spec)
    grads = tr.backward(blob, spec, torch.from_numpy(g["grad_enh"]).cuda()).cpu().numpy()
    def worst(a):
        rows = []
        for name, numel, off in L.param_table():
            t = t64[off:off + numel]
            if not t.any():
                continue
            sc = max(np.abs(t).max(), 1e-3 * np.abs(t64).max())
            rows.append((float(np.abs(a[off:off + numel] - t).max() / sc), name))
        rows.sort(reverse=True)
        return rows[:3]
    print(tag, "HIP vs fp64:", worst(grads))
    print(tag, "reference golden vs fp64:", worst(g["grads"]))
