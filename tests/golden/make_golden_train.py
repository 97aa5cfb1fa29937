#!/usr/bin/env python3
"""Generates the train-step fixtures by IMPORTING the reference (this container only).

    python tests/golden/make_golden_train.py

Writes, for the shipped checkpoint ("dns3") and for the randomised model ("rand", same
randomise() seed as make_golden.py so it equals tests/golden/params_rand.f32):

  trainstep_<tag>_B3_T12.npz
      noisy_spec, clean_spec      (3,257,12,2)  Hann-window STFTs (train.py:247-263)
      enh                          model(noisy_spec) in train mode (batch-statistics BatchNorm)
      loss                         HybridLoss (loss.py:30-71)
      grad_enh                     d loss / d enh   (what the model's backward receives)
      grads                        flat fp32, canonical blob layout (params_manifest.json offsets);
                                   entries of buffers (BatchNorm running statistics, ERB bank) are 0
      params_after                 canonical blob after the forward: running statistics updated
                                   (momentum 0.1, unbiased variance), everything else unchanged
      stage:<name>                 train-mode activations at a few stage boundaries, (B,C,T,F)
      grad_norm                    clip_grad_norm_(.., 3.0) return value (train.py:282-284)

Fixtures are data only; this script is the generator and is never run by the tests.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402  (imports the reference)

OUT = MG.OUT


def flat_blob(sd):
    return np.concatenate([v.detach().cpu().numpy().astype(np.float32).ravel() for _, v in MG.canonical_items(sd)])


def run(tag, model, seed):
    model.train()
    g = torch.Generator().manual_seed(seed)
    B, L = 3, 256 * 11
    clean = torch.randn(B, L, generator=g) * 0.05
    noisy = clean + torch.randn(B, L, generator=g) * 0.05
    hann = torch.hann_window(512)
    ns, cs = MG.stft(noisy, hann), MG.stft(clean, hann)
    stages = {}
    hooks = []

    def keep(name):
        def fn(_m, _i, o):
            stages[name] = (o[0] if isinstance(o, tuple) else o).detach().numpy().copy()
        return fn
    for name, mod in (("en0", model.encoder.en_convs[0]), ("en1", model.encoder.en_convs[1]),
                      ("en2", model.encoder.en_convs[2]), ("en4", model.encoder.en_convs[4]),
                      ("gtcn1", model.gtcn1), ("gtcn2", model.gtcn2),
                      ("de0", model.decoder.de_convs[0]), ("de2", model.decoder.de_convs[2]),
                      ("de3", model.decoder.de_convs[3]), ("de4", model.decoder.de_convs[4])):
        hooks.append(mod.register_forward_hook(keep(name)))
    enh = model(ns)
    enh.retain_grad()
    for h in hooks:
        h.remove()
    loss = MG.HybridLoss(512, 256, 512, 512)(enh, cs)
    loss.backward()
    named = dict(model.named_parameters())
    parts = []
    for k, v in MG.canonical_items(model.state_dict()):
        p = named.get(k)
        gk = p.grad if (p is not None and p.grad is not None) else torch.zeros_like(v, dtype=torch.float32)
        parts.append(gk.detach().numpy().astype(np.float32).ravel())
    grads = np.concatenate(parts)
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 3.0)
    np.savez_compressed(
        os.path.join(OUT, f"trainstep_{tag}_B3_T12.npz"), noisy_spec=ns.numpy(), clean_spec=cs.numpy(),
        enh=enh.detach().numpy(), loss=float(loss), grad_enh=enh.grad.numpy(), grads=grads,
        params_after=flat_blob(model.state_dict()), grad_norm=float(gn),
        **{"stage:" + k: v for k, v in stages.items()})
    n_train = int(sum(p.numel() for p in model.parameters() if p.requires_grad))
    return {"loss": float(loss), "grad_norm": float(gn), "n_trainable": n_train,
            "n_trainable_tensors": int(sum(1 for p in model.parameters() if p.requires_grad))}


def main():
    ck = torch.load(os.path.join(MG.REF, "gtcrn_micro/ckpts/best_model_dns3.tar"), map_location="cpu",
                    weights_only=False)
    meta = {"torch": torch.__version__}
    m = MG.GTCRNMicro()
    m.load_state_dict(ck["model"])
    assert np.array_equal(flat_blob(m.state_dict()), np.fromfile(os.path.join(OUT, "params_dns3.f32"), np.float32))
    meta["dns3"] = run("dns3", m, 21)
    torch.manual_seed(1234)
    m = MG.GTCRNMicro()
    MG.randomise(m, 99)
    assert np.array_equal(flat_blob(m.state_dict()), np.fromfile(os.path.join(OUT, "params_rand.f32"), np.float32))
    meta["rand"] = run("rand", m, 22)
    json.dump(meta, open(os.path.join(OUT, "MANIFEST_train.json"), "w"), indent=1)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
