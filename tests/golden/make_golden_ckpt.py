#!/usr/bin/env python3
"""Generates the checkpoint fixture by IMPORTING the reference (this container only).

    python tests/golden/make_golden_ckpt.py

A seed-fixed reference model takes two reference train steps on the CPU (Hann STFT, HybridLoss, clip 3.0, Adam,
warm-up-cosine schedule: train.py:239-288), then the reference's OWN ``Trainer._save_checkpoint`` (train.py:200-221)
writes ``ckpt_ref/model_002.tar`` -- the {"epoch","optimizer","scheduler","model"} dict with the 388-key state_dict.
Next to it: ``ckpt_ref_io.npz`` = a probe spectrogram and the reference model's eval-mode output for it, plus the
learning rate and Adam step count stored in the checkpoint.  Fixtures are data; this script is never run by tests.

train.py imports packages that are absent here (soundfile, omegaconf, pesq, tensorboard, joblib is present); they
are only needed by its data/eval loop, so empty stand-in modules satisfy the import -- none of their code runs.
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402  (puts /root/reference on the path, stubs soundfile)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)


_stub("omegaconf", OmegaConf=object)
_stub("pesq", pesq=lambda *a, **k: 0.0)
_stub("torch.utils.tensorboard", SummaryWriter=object)
_stub("librosa")
_stub("pystoi", stoi=lambda *a, **k: 0.0)
_stub("p_tqdm")

import gtcrn_micro.train as RT  # noqa: E402

OUT = MG.OUT


def main():
    torch.manual_seed(1234)
    model = RT.Model()
    MG.randomise(model, 77)                     # non-trivial BatchNorm statistics / PReLU slopes
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    sched = RT.WarmupLR(opt, 25000, 250000, 1e-3, 1e-6)
    loss_func = RT.Loss()
    g = torch.Generator().manual_seed(5)
    hann = torch.hann_window(512)
    model.train()
    for _ in range(2):                          # two iterations of Trainer._train_epoch (train.py:244-288)
        clean = torch.randn(2, 256 * 9, generator=g) * 0.05
        noisy = clean + torch.randn(2, 256 * 9, generator=g) * 0.05
        ns, cs = MG.stft(noisy, hann), MG.stft(clean, hann)
        loss = loss_func(model(ns), cs)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 3.0)
        opt.step()
        sched.step()
    # the reference's own writer, on a Trainer shell that holds exactly what _save_checkpoint reads
    tr = object.__new__(RT.Trainer)
    tr.model, tr.optimizer, tr.scheduler = model, opt, sched
    tr.world_size, tr.best_score = 1, 0.0
    tr.checkpoint_path = os.path.join(OUT, "ckpt_ref")
    os.makedirs(tr.checkpoint_path, exist_ok=True)
    tr._save_checkpoint(2, -1.0)
    model.eval()
    probe = torch.randn(2, 257, 9, 2, generator=g) * 0.3
    with torch.no_grad():
        out = model(probe)
    np.savez_compressed(os.path.join(OUT, "ckpt_ref_io.npz"), probe=probe.numpy(), out=out.numpy(),
                        lr=np.float64(opt.param_groups[0]["lr"]),
                        adam_step=np.float64(next(iter(opt.state.values()))["step"]),
                        n_keys=np.int64(len(model.state_dict())))
    print("wrote", os.listdir(tr.checkpoint_path), os.path.getsize(os.path.join(tr.checkpoint_path, "model_002.tar")))


if __name__ == "__main__":
    main()
