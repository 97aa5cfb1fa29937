"""numpy emulation of the kernels' slot-space dataflow, driven by the PACKED buffers the HIP
library produces (gtcrn_pack_params_host).  Test infrastructure: it lets the CPU suite check
the packer (BatchNorm folding, ConvTranspose weight contract, shuffle-as-renaming, skip index
tables) against the oracle without a GPU.  Offline only (zero history), one utterance."""
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def layout_constants():
    """Evaluates the `constexpr int NAME = expr;` lines of csrc/layout.h."""
    txt = open(os.path.join(_HERE, "..", "gtcrn_micro_amd", "csrc", "layout.h")).read()
    env = {}
    for name, expr in re.findall(r"constexpr int (\w+) = ([^;]+);", txt):
        env[name] = int(eval(expr, {}, env))
    return env


K = layout_constants()


def prelu(v, a):
    return np.where(v >= 0, v, np.float32(a) * v).astype(np.float32)


def _mat(buf, off):
    """16x16 slot matrix at float offset `off`; the packer stores rows 4..7 and 12..15 with their two 8-float halves
    exchanged (bank-conflict-free A-fragment reads, csrc/pack.cpp swizzle_matrix): undone here."""
    m = buf[off:off + 256].reshape(16, 16).copy()
    rows = [n for n in range(16) if (n >> 2) & 1]
    m[rows] = np.concatenate([m[rows, 8:], m[rows, :8]], axis=1)
    return m


def gtconv(x, pb, ib, dense):
    """x (T,33,16 slots) -> (T,33,16 slots); mirrors gtconv_block<DENSE> in kernels.hip."""
    T = x.shape[0]
    a1, a2 = pb[K["GB_SLOPE"]], pb[K["GB_SLOPE"] + 1]
    h = prelu(x @ _mat(pb, K["GB_PC1_A"]).T + pb[K["GB_PC1_B"]:K["GB_PC1_B"] + 16], a1)
    hp = np.zeros((T + 2, 35, 16), np.float32)
    hp[2:, 1:34] = h
    acc = np.broadcast_to(pb[K["GB_DW_B"]:K["GB_DW_B"] + 16], (T, 33, 16)).astype(np.float32).copy()
    for kt in range(3):
        for kf in range(3):
            if dense:   # tap (t-kt, f+1-kf)
                tap = hp[2 - kt:2 - kt + T, 2 - kf:2 - kf + 33]
                acc += tap @ _mat(pb, K["GB_DN_A"] + (kt * 3 + kf) * 256).T
            else:       # tap (t-2+kt, f-1+kf)
                tap = hp[kt:kt + T, kf:kf + 33]
                acc += pb[K["GB_DW_W"] + (kt * 3 + kf) * 16:K["GB_DW_W"] + (kt * 3 + kf) * 16 + 16] * tap
    hd = prelu(acc, a2)
    keep = pb[K["GB_KEEP"]:K["GB_KEEP"] + 16]
    v = keep * x + pb[K["GB_PC2_B"]:K["GB_PC2_B"] + 16] + hd @ _mat(pb, K["GB_PC2_A"]).T
    slot_of_c, x2slots = ib[:8], ib[8:16]
    e = (v[:, :, slot_of_c] ** 2).sum(axis=1) / np.float32(33.0)          # (T,8)
    ep = np.concatenate([np.zeros((2, 8), np.float32), e])
    dw = pb[K["GB_TRA_DW"]:K["GB_TRA_DW"] + 24].reshape(8, 3)
    y = pb[K["GB_TRA_DB"]:K["GB_TRA_DB"] + 8] + sum(dw[:, k] * ep[k:k + T] for k in range(3))
    z = pb[K["GB_TRA_PB"]:K["GB_TRA_PB"] + 8] + y @ pb[K["GB_TRA_PW"]:K["GB_TRA_PW"] + 64].reshape(8, 8).T
    G = np.ones((T, 16), np.float32)
    G[:, slot_of_c] = 1.0 / (1.0 + np.exp(-z))
    assert sorted(list(slot_of_c) + list(x2slots)) == list(range(16))
    return (v * G[:, None, :]).astype(np.float32)


def tcn(x, pk, d):
    T = x.shape[0]
    a1, a2, a3 = pk[K["TCN_SLOPE"]:K["TCN_SLOPE"] + 3]
    y1 = prelu(x @ _mat(pk, K["TCN_A1"]).T + pk[K["TCN_B1"]:K["TCN_B1"] + 16], a1)
    yp = np.concatenate([np.zeros((2 * d, 33, 16), np.float32), y1])
    w = pk[K["TCN_DW"]:K["TCN_DW"] + 48].reshape(3, 16)
    y2 = pk[K["TCN_B2"]:K["TCN_B2"] + 16] + sum(w[k] * yp[k * d:k * d + T] for k in range(3))
    y2 = prelu(y2, a2)
    y3 = y2 @ _mat(pk, K["TCN_A3"]).T + pk[K["TCN_B3"]:K["TCN_B3"] + 16]
    return prelu(y3 + x, a3)


def forward(F, I, spec):
    """spec (257,T,2) -> dict of stage tensors in slot order + 'out' (257,T,2)."""
    T = spec.shape[1]
    rec = {}
    re_, im_ = spec[..., 0].T, spec[..., 1].T
    feat = np.stack([np.sqrt(re_ * re_ + im_ * im_ + np.float32(1e-12)), re_, im_]).astype(np.float32)
    E, D = F[K["P_ENC"]:], F[K["P_DEC"]:]
    eb = np.zeros((3, T, 129), np.float32)
    eb[:, :, :65] = feat[:, :, :65]
    for j in range(64):
        lo, n = I[K["I_ERB_LO"] + j], I[K["I_ERB_N"] + j]
        w = E[K["E_ERB_W"] + j * K["ERB_MAXBW"]:K["E_ERB_W"] + j * K["ERB_MAXBW"] + n]
        eb[:, :, 65 + j] = (feat[:, :, 65 + lo:65 + lo + n] * w).sum(-1)
    sw = E[K["E_SFE_W"]:K["E_SFE_W"] + 9].reshape(3, 3)
    ebp = np.pad(eb, ((0, 0), (0, 0), (1, 1)))
    f0 = sum(sw[:, k, None, None] * ebp[:, :, k:k + 129] for k in range(3)).astype(np.float32)
    # en0: im2col column e = c*5+k, input bin 2fo-2+k
    f0p = np.pad(f0, ((0, 0), (0, 0), (2, 2)))
    cols = np.zeros((T, 65, 16), np.float32)
    for e in range(15):
        cols[:, :, e] = f0p[e // 5][:, (e % 5) + 2 * np.arange(65)]
    x0 = prelu(cols @ _mat(E, K["E_EN0_A"]).T + E[K["E_EN0_B"]:K["E_EN0_B"] + 16], E[K["E_EN0_S"]])
    rec["en0"] = x0
    x0p = np.pad(x0, ((0, 0), (2, 2), (0, 0)))
    acc = np.broadcast_to(E[K["E_EN1_B"]:K["E_EN1_B"] + 16], (T, 33, 16)).astype(np.float32).copy()
    for k in range(5):
        acc += x0p[:, k + 2 * np.arange(33)] @ _mat(E, K["E_EN1_A"] + k * 256).T
    x = prelu(acc, E[K["E_EN1_S"]])

    def stored(q, v):   # en1..en3 are stored in their decoder consumer's slot order (I_ENST)
        out = np.empty_like(v)
        out[:, :, I[K["I_ENST"] + q * 16:K["I_ENST"] + q * 16 + 16]] = v
        return out

    rec["en1"] = stored(0, x)
    for k in range(3):
        x = gtconv(x, E[K["E_BLK"] + k * K["GB_SIZE"]:], I[K["I_ENC_BLK"] + k * 16:], False)
        rec[f"en{2 + k}"] = stored(k + 1, x) if k < 2 else x
    for g in range(2):
        for k in range(4):
            x = tcn(x, F[K["P_GTCN"] + g * K["GTCN_SIZE"] + k * K["TCN_SIZE"]:], 1 << k)
        rec[f"gtcn{g + 1}"] = x
    x = x + rec["en4"]
    for j in range(3):
        x = gtconv(x, D[K["D_BLK"] + j * K["GBD_SIZE"]:], I[K["I_DEC_BLK"] + j * 16:], True)
        rec[f"de{j}"] = x
        x = x + rec[f"en{3 - j}"]
    # de3 gather form
    xp = np.pad(x, ((0, 0), (1, 1), (0, 0)))           # index f+1
    b3 = D[K["D_DE3_B"]:K["D_DE3_B"] + 16]
    ye = b3 + xp[:, 2:35] @ _mat(D, K["D_DE3_AE"]).T + x @ _mat(D, K["D_DE3_AE"] + 256).T + \
        xp[:, 0:33] @ _mat(D, K["D_DE3_AE"] + 512).T
    yo = b3 + xp[:, 2:35] @ _mat(D, K["D_DE3_AO"]).T + x @ _mat(D, K["D_DE3_AO"] + 256).T
    y = np.zeros((T, 65, 16), np.float32)
    y[:, 0::2] = ye
    y[:, 1::2] = yo[:, :32]
    y = prelu(y, D[K["D_DE3_S"]])
    rec["de3"] = y
    y = y + rec["en0"]
    z = y @ _mat(D, K["D_DE4_A"]).T                      # (T,65,16): row o*5+k
    m = np.zeros((2, T, 129), np.float32)
    for o in range(2):
        m[o] = D[K["D_DE4_B"] + o]
        for k in range(5):
            fo = 2 * np.arange(65) - 2 + k
            ok = (fo >= 0) & (fo < 129)
            m[o][:, fo[ok]] += z[:, ok, o * 5 + k]
    m = np.tanh(m)
    rec["de4"] = m
    mm = np.zeros((2, T, 257), np.float32)
    mm[:, :, :65] = m[:, :, :65]
    for i in range(192):
        lo, n = I[K["I_BS_LO"] + i], I[K["I_BS_N"] + i]
        w = D[K["D_BS_W"] + i * K["ERB_MAXBS"]:K["D_BS_W"] + i * K["ERB_MAXBS"] + n]
        mm[:, :, 65 + i] = (m[:, :, 65 + lo:65 + lo + n] * w).sum(-1)
    out = np.empty_like(spec)
    out[..., 0] = (re_ * mm[0] - im_ * mm[1]).T
    out[..., 1] = (im_ * mm[0] + re_ * mm[1]).T
    rec["out"] = out
    return rec


PERM_INDEX = {"en0": 0, "en1": 1, "en2": 2, "en3": 3, "en4": 4, "gtcn1": 4, "gtcn2": 4,
              "de0": 5, "de1": 6, "de2": 7, "de3": 8}


def to_logical(name, x, I):
    """(T,F,16 slots) -> (16 logical channels, T, F)."""
    perm = I[K["I_PERM"] + PERM_INDEX[name] * 16:K["I_PERM"] + PERM_INDEX[name] * 16 + 16]
    out = np.empty((16,) + x.shape[:2], np.float32)
    out[perm] = np.transpose(x, (2, 0, 1))
    return out
