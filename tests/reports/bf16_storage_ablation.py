#!/usr/bin/env python3
"""Diagnostic (CPU, a few minutes): WHERE does the gradient error of bf16-stored activations come from?

The HIP trainer's bf16 storage rounds every tensor the backward re-reads AND lets the forward chain consume the rounded
copies (the next conv reads the stored bf16 activation).  Round 2 measured a 0.24-0.36 relative-L2 gradient error
against the float64 graph, independent of the batch size.  This script separates the two effects on the PyTorch-CPU
port of the same graph (oracle/torch_port.py), for a fixed upstream gradient like tests/test_gpu_train.py:

  save-only   the forward is exact fp32; only what autograd SAVES for the backward is rounded to bf16
              (torch.autograd.graph.saved_tensors_hooks), all saved tensors or one op class at a time;
  chain       the forward chain itself consumes rounded activations (straight-through rounding after every
              activation / block output), the saved tensors are those rounded values: the round-2 design.

    python tests/reports/bf16_storage_ablation.py            # prints a table, writes profiles/r03_bf16_ablation.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.torch_port as TP   # noqa: E402

REAL_F = torch.nn.functional
OPS = ("conv2d", "conv_transpose2d", "batch_norm", "prelu", "conv1d", "linear")


def rnd(t, fmt):
    if fmt == "bf16":
        return t.to(torch.bfloat16).to(t.dtype)
    if fmt == "fp16":
        return t.to(torch.float16).to(t.dtype)
    return t


class RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fmt):
        return rnd(x, fmt)

    @staticmethod
    def backward(ctx, g):
        return g, None


class FShim:
    """torch.nn.functional with per-op-class rounding of what the op saves for its backward."""

    def __init__(self, policy, params, chain=None):
        self.policy, self.param_ids, self.chain = policy, {id(p) for p in params}, chain

    def _pack(self, op):
        fmt = self.policy.get(op)

        def pack(t):
            if fmt and t.is_floating_point() and t.numel() >= 4096 and id(t) not in self.param_ids:
                return rnd(t.detach(), fmt)
            return t
        return pack

    def __getattr__(self, name):
        real = getattr(REAL_F, name)
        if name not in OPS:
            return real

        def wrapped(*a, **k):
            with torch.autograd.graph.saved_tensors_hooks(self._pack(name), lambda t: t):
                out = real(*a, **k)
            if self.chain and name == "prelu":                  # the forward chain consumes the stored activation
                out = RoundSTE.apply(out, self.chain)
            return out
        return wrapped


def grads(blob, spec, gout, dtype, policy=None, chain=None, other=None):
    port = TP.TorchPort(blob.copy(), train=True, dtype=dtype)
    TP.F = FShim(policy or {}, port.p.values(), chain)
    try:
        fmt_other = other

        def pack_other(t):
            if fmt_other and t.is_floating_point() and t.numel() >= 4096 and not t.requires_grad:
                return rnd(t.detach(), fmt_other)
            return t
        with torch.autograd.graph.saved_tensors_hooks(pack_other, lambda t: t):
            enh, g = port.backward_from(spec, gout)
    finally:
        TP.F = REAL_F
    return enh, g.astype(np.float64)


def err(g, ref):
    m = ref != 0
    return (float(np.linalg.norm(g[m] - ref[m]) / np.linalg.norm(ref[m])),
            float(np.dot(g[m], ref[m]) / (np.linalg.norm(g[m]) * np.linalg.norm(ref[m]))))


def main():
    torch.set_num_threads(8)
    rows = []
    for tag in ("rand", "dns3"):
        blob = np.fromfile(os.path.join(ROOT, "tests", "golden", f"params_{tag}.f32"), np.float32)
        for B, T in ((3, 12), (8, 40)):
            rng = np.random.default_rng(7)
            spec = (rng.standard_normal((B, 257, T, 2)) * 0.3).astype(np.float32)
            gout = (rng.standard_normal((B, 257, T, 2)) * 0.1).astype(np.float32)
            e64, ref = grads(blob, spec, gout, torch.float64)
            cases = [("fp32 (no rounding)", {}, None, None),
                     ("save-only: everything bf16", {o: "bf16" for o in OPS}, None, "bf16")]
            cases += [(f"save-only: only {o} bf16", {o: "bf16"}, None, None) for o in OPS]
            cases += [("save-only: only the other ops (mul, tanh, ...) bf16", {}, None, "bf16"),
                      ("save-only: all but batch_norm bf16", {o: "bf16" for o in OPS if o != "batch_norm"}, None, "bf16"),
                      ("save-only: batch_norm fp16, rest bf16", dict({o: "bf16" for o in OPS}, batch_norm="fp16"), None, "bf16"),
                      ("chain: forward consumes bf16 activations (round-2 design)", {o: "bf16" for o in OPS}, "bf16", "bf16"),
                      ("chain only: rounded forward chain, fp32 saves", {}, "bf16", None)]
            for name, pol, chain, other in cases:
                enh, g = grads(blob, spec, gout, torch.float32, pol, chain, other)
                l2, cos = err(g, ref)
                fwd = float(np.linalg.norm(enh - e64) / np.linalg.norm(e64))
                rows.append({"weights": tag, "B": B, "T": T, "case": name, "grad_rel_l2": l2, "grad_cosine": cos,
                             "forward_rel_l2": fwd})
                print(f"{tag} B={B} T={T:3d}  {name:62s} grad rel-L2 {l2:9.2e}  cos {cos:.5f}  fwd {fwd:.1e}", flush=True)
    out = os.path.join(ROOT, "profiles", "r03_bf16_ablation.json")
    json.dump(rows, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
