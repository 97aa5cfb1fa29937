#!/usr/bin/env python3
"""Generate the committed golden vectors by IMPORTING the reference (PyTorch CPU).

Run only in the build container (the reference does not travel to the GPU box):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

Everything written here is DATA (inputs + expected outputs + the shipped
checkpoint's tensors as a raw fp32 blob); no reference source is stored.
The torch version used is recorded in MANIFEST.json because fp32 conv/FFT
results can differ at ~1e-6 between torch builds.

Fixtures (all small):
  params_dns3.f32 / params_rand.f32   flat fp32 parameter blobs, canonical order
  params_manifest.json                [name, shape, offset] of the canonical order
  offline_{dns3,rand}_T17.npz         1 clip x 4096 samples: every stage boundary
  stream_{dns3,rand}_T17.npz          same clip frame by frame + caches after 0,1,16
  example_noisy1_head.npz             first 2 s of examples/noisy1.wav + enh1.wav (int16)
  conv_wrappers.npz                   StreamConv2d/StreamConvTranspose2d known answers
  train_dns3_B4_T17.npz               one train-mode fwd/bwd (Hann window, HybridLoss)
  causality_T126.npz                  the reference's own causality test, frozen
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
sys.path.insert(0, REF)
sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))  # only used in __main__

from gtcrn_micro.models.gtcrn_micro import GTCRNMicro  # noqa: E402
from gtcrn_micro.streaming.gtcrn_micro_stream import StreamGTCRNMicro  # noqa: E402
from gtcrn_micro.streaming.conversion.convert import convert_to_stream  # noqa: E402
from gtcrn_micro.streaming.conversion.convolution import (  # noqa: E402
    StreamConv2d,
    StreamConvTranspose2d,
)
from gtcrn_micro.loss import HybridLoss  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)
torch.use_deterministic_algorithms(True)


def canonical_items(sd):
    """Canonical parameter order = state_dict order minus int64 num_batches_tracked."""
    return [(k, v) for k, v in sd.items() if not k.endswith("num_batches_tracked")]


def write_blob(sd, fname):
    items = canonical_items(sd)
    flat = np.concatenate([v.detach().cpu().numpy().astype(np.float32).ravel() for _, v in items])
    flat.tofile(os.path.join(OUT, fname))
    manifest, off = [], 0
    for k, v in items:
        manifest.append([k, list(v.shape), off])
        off += v.numel()
    return manifest, off


def randomise(model, seed):
    """Random weights incl. non-trivial BN statistics and PReLU slopes of both signs."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, mod in model.named_modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
            elif isinstance(mod, nn.PReLU):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) * 0.9 - 0.3)
            elif isinstance(mod, (nn.Conv2d, nn.ConvTranspose2d, nn.Conv1d)):
                fan = mod.weight[0].numel() if not isinstance(mod, nn.ConvTranspose2d) else (
                    mod.weight.shape[0] * mod.weight.shape[2] * mod.weight.shape[3])
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) / np.sqrt(max(fan, 1)))
                if mod.bias is not None:
                    mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
    # ERB bank stays the fixed (frozen) filterbank.


def sqrt_hann():
    return torch.hann_window(512).pow(0.5)


def stft(x, win):
    return torch.stft(x, 512, 256, 512, win, return_complex=False)


def istft(y, win):
    return torch.istft(torch.view_as_complex(y.contiguous()), 512, 256, 512, win)


def offline_stages(model, wave):
    """Run model(spec) while recording every stage boundary of SURVEY.md section 2a."""
    rec = {}
    hooks = []

    def save(name):
        def fn(_m, _i, o):
            rec[name] = (o[0] if isinstance(o, tuple) else o).detach().numpy().copy()
        return fn

    hooks.append(model.sfe.register_forward_hook(save("sfe")))
    for i, m in enumerate(model.encoder.en_convs):
        hooks.append(m.register_forward_hook(save(f"en{i}")))
    for gi, g in enumerate((model.gtcn1, model.gtcn2)):
        for bi, b in enumerate(g.blocks):
            hooks.append(b.register_forward_hook(save(f"gtcn{gi + 1}_b{bi}")))
    for i, m in enumerate(model.decoder.de_convs):
        hooks.append(m.register_forward_hook(save(f"de{i}")))
    bm0, bs0 = model.erb.bm, model.erb.bs

    def bm(x):
        rec["feat"] = x.detach().numpy().copy()
        y = bm0(x)
        rec["erb_bm"] = y.detach().numpy().copy()
        return y

    def bs(x):
        y = bs0(x)
        rec["erb_bs"] = y.detach().numpy().copy()
        return y

    model.erb.bm, model.erb.bs = bm, bs
    win = sqrt_hann()
    spec = stft(wave, win)  # (B?,257,T,2)
    if spec.dim() == 3:
        spec = spec[None]
    with torch.inference_mode():
        out = model(spec)
    model.erb.bm, model.erb.bs = bm0, bs0
    for h in hooks:
        h.remove()
    wave_out = istft(out[0], win)
    rec.update(wave=wave.numpy().copy(), spec=spec.numpy().copy(), spec_enh=out.numpy().copy(),
               wave_out=wave_out.numpy().copy(), window=win.numpy().copy())
    return rec


def stream_run(model, spec, snap_frames=(0, 1, 16)):
    sm = StreamGTCRNMicro().eval()
    convert_to_stream(sm, model)
    B = spec.shape[0]
    conv_cache = torch.zeros(2, B, 16, 6, 33)
    tra_cache = torch.zeros(2, 3, B, 8, 2)
    tcn_cache = [[torch.zeros(B, 16, 2 * d, 33) for d in (1, 2, 4, 8)] for _ in range(2)]
    ys, rec = [], {}
    with torch.no_grad():
        for i in range(spec.shape[2]):
            y, conv_cache, tra_cache, tcn_cache = sm(spec[:, :, i:i + 1], conv_cache, tra_cache, tcn_cache)
            ys.append(y.numpy().copy())
            if i in snap_frames:
                rec[f"conv_cache_f{i}"] = conv_cache.numpy().copy()
                rec[f"tra_cache_f{i}"] = tra_cache.numpy().copy()
                for g in range(2):
                    for k in range(4):
                        rec[f"tcn_cache_f{i}_g{g}_b{k}"] = tcn_cache[g][k].numpy().copy()
    rec["spec"] = spec.numpy().copy()
    rec["spec_enh_stream"] = np.concatenate(ys, axis=2)
    return rec


def main():
    meta = {"torch": torch.__version__, "numpy": np.__version__, "threads": 1}

    # ---- models -----------------------------------------------------------
    ck = torch.load(os.path.join(REF, "gtcrn_micro/ckpts/best_model_dns3.tar"),
                    map_location="cpu", weights_only=False)
    m_dns = GTCRNMicro().eval()
    m_dns.load_state_dict(ck["model"])
    torch.manual_seed(1234)
    m_rnd = GTCRNMicro().eval()
    randomise(m_rnd, 99)

    man, n = write_blob(m_dns.state_dict(), "params_dns3.f32")
    man2, n2 = write_blob(m_rnd.state_dict(), "params_rand.f32")
    assert man == man2 and n == n2
    json.dump({"n_floats": n, "tensors": man}, open(os.path.join(OUT, "params_manifest.json"), "w"))
    meta["n_param_floats"] = n

    # ---- (1)/(3) offline stage vectors, T = 17 ----------------------------
    g = torch.Generator().manual_seed(43)  # the reference's seed, train.py:27
    wave = (torch.randn(4096, generator=g) * 0.1)
    for tag, m in (("dns3", m_dns), ("rand", m_rnd)):
        rec = offline_stages(m, wave)
        np.savez_compressed(os.path.join(OUT, f"offline_{tag}_T17.npz"), **rec)
        srec = stream_run(m, torch.from_numpy(rec["spec"]))
        srec["spec_enh_offline"] = rec["spec_enh"]
        np.savez_compressed(os.path.join(OUT, f"stream_{tag}_T17.npz"), **srec)
        meta[f"stream_vs_offline_maxabs_{tag}"] = float(
            np.abs(srec["spec_enh_stream"] - rec["spec_enh"]).max())

    # small batched case (B=3, ragged lengths are the caller's business; here equal lengths)
    wb = torch.randn(3, 2048 + 256 * 3, generator=g) * 0.2
    spec_b = stft(wb, sqrt_hann())
    with torch.inference_mode():
        out_b = m_dns(spec_b)
    np.savez_compressed(os.path.join(OUT, "offline_dns3_B3_T12.npz"), wave=wb.numpy(), spec=spec_b.numpy(),
                        spec_enh=out_b.numpy(),
                        wave_out=torch.stack([istft(o, sqrt_hann()) for o in out_b]).numpy())

    # ---- (5) known-answer: shipped example pair ----------------------------
    from scipy.io import wavfile
    fs, noisy = wavfile.read(os.path.join(REF, "gtcrn_micro/examples/gtcrn_micro/noisy1.wav"))
    fs2, enh = wavfile.read(os.path.join(REF, "gtcrn_micro/examples/gtcrn_micro/enh1.wav"))
    assert fs == fs2 == 16000 and noisy.dtype == np.int16
    NH = 32000
    np.savez_compressed(os.path.join(OUT, "example_noisy1_head.npz"), noisy=noisy[:NH], enh=enh[:NH - 1024],
                        full_len_noisy=len(noisy), full_len_enh=len(enh))
    # sanity: reference reproduces its own example on the head
    x = torch.from_numpy(noisy[:NH].astype(np.float32) / 32768.0)
    with torch.inference_mode():
        y = istft(m_dns(stft(x, sqrt_hann())[None])[0], sqrt_hann()).numpy()
    meta["example_head_maxabs_lsb"] = float(np.abs(y[:NH - 1024] * 32768.0 - enh[:NH - 1024]).max())

    # ---- conv wrappers (reference tests/streaming/conversion/test_convolution.py) ----
    torch.manual_seed(7)
    rec = {}
    conv = nn.Conv2d(1, 1, 3)
    x = torch.randn(1, 1, 10, 6)
    with torch.no_grad():
        rec["c2d_w"], rec["c2d_b"], rec["c2d_x"] = conv.weight.numpy(), conv.bias.numpy(), x.numpy()
        rec["c2d_y"] = conv(nn.functional.pad(x, [0, 0, 2, 0])).numpy()
    kt, dt = 3, 2
    pt = (kt - 1) * dt
    de = nn.ConvTranspose2d(4, 8, (kt, 1), stride=(1, 1), padding=(pt, 1), dilation=(dt, 2), groups=1)
    sde = StreamConvTranspose2d(4, 8, (kt, 1), stride=(1, 1), padding=(0, 1), dilation=(dt, 2), groups=1)
    convert_to_stream(sde, de)
    x = torch.randn(1, 4, 100, 6)
    with torch.no_grad():
        rec["ct2d_w"], rec["ct2d_b"], rec["ct2d_x"] = de.weight.numpy(), de.bias.numpy(), x.numpy()
        rec["ct2d_y"] = de(nn.functional.pad(x, [0, 0, pt, 0])).numpy()
        rec["ct2d_w_stream"] = sde.ConvTranspose2d.weight.numpy()
        # stream step by step as the reference test does
        cache = torch.zeros(1, 4, pt, 6)
        outs = []
        for i in range(100):
            o, cache = sde(x[:, :, i:i + 1], cache)
            outs.append(o)
        rec["ct2d_y_stream"] = torch.cat(outs, dim=2).numpy()
    # the model's own decoder shape: dense 16->16 (3,3) transposed, pad (0,1)
    de2 = nn.ConvTranspose2d(16, 16, (3, 3), padding=(0, 1))
    x = torch.randn(2, 16, 9, 33)
    with torch.no_grad():
        rec["ct33_w"], rec["ct33_b"], rec["ct33_x"] = de2.weight.numpy(), de2.bias.numpy(), x.numpy()
        rec["ct33_y"] = de2(x).numpy()  # T+2 frames
    np.savez_compressed(os.path.join(OUT, "conv_wrappers.npz"), **rec)

    # ---- causality test of the reference, frozen (tests/models/test_gtcrn_micro.py) ----
    g2 = torch.Generator().manual_seed(5)
    a, b, c = (torch.randn(1, 16000, generator=g2) for _ in range(3))
    x1, x2 = torch.cat([a, b], 1), torch.cat([a, c], 1)
    with torch.inference_mode():
        y1 = istft(m_rnd(stft(x1, sqrt_hann()))[0], sqrt_hann())
        y2 = istft(m_rnd(stft(x2, sqrt_hann()))[0], sqrt_hann())
    meta["causality_prefix_maxabs"] = float((y1[:16000 - 512] - y2[:16000 - 512]).abs().max())
    meta["causality_suffix_maxabs"] = float((y1[16000:] - y2[16000:]).abs().max())
    np.savez_compressed(os.path.join(OUT, "causality_T126.npz"), x1=x1.numpy(), x2=x2.numpy(),
                        y1=y1.numpy(), y2=y2.numpy())

    # ---- (4) one train-mode step: Hann window (train.py:247-263), HybridLoss ----------
    torch.manual_seed(43)
    m_tr = GTCRNMicro()
    m_tr.load_state_dict(ck["model"])
    m_tr.train()
    g3 = torch.Generator().manual_seed(11)
    clean = torch.randn(4, 4096, generator=g3) * 0.05
    noisy_t = clean + torch.randn(4, 4096, generator=g3) * 0.05
    hann = torch.hann_window(512)
    ns, cs = stft(noisy_t, hann), stft(clean, hann)
    enh_t = m_tr(ns)
    loss = HybridLoss(512, 256, 512, 512)(enh_t, cs)
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(m_tr.parameters(), 3.0)
    grads = {k: p.grad.numpy().copy() for k, p in m_tr.named_parameters() if p.grad is not None}
    keep = ["encoder.en_convs.0.conv.weight", "encoder.en_convs.2.tra.point_conv.weight",
            "gtcn1.blocks.3.conv2.weight", "gtcn2.blocks.0.act3.weight",
            "decoder.de_convs.0.depth_conv.weight", "decoder.de_convs.4.conv.bias"]
    np.savez_compressed(
        os.path.join(OUT, "train_dns3_B4_T17.npz"), noisy=noisy_t.numpy(), clean=clean.numpy(),
        noisy_spec=ns.numpy(), clean_spec=cs.numpy(), enh=enh_t.detach().numpy(), loss=float(loss),
        grad_norm=float(gn),
        bn0_running_mean=m_tr.encoder.en_convs[0].bn.running_mean.numpy(),
        bn0_running_var=m_tr.encoder.en_convs[0].bn.running_var.numpy(),
        **{"grad_clipped:" + k: grads[k] for k in keep})
    meta["train_loss"] = float(loss)
    meta["train_grad_norm"] = float(gn)

    json.dump(meta, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
