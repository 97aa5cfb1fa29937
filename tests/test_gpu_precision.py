"""What keeps the headline "fp32-class": the HIP forward against the SAME graph in float64, the device-side exact bf16
split the dense 3x3 runs on, the reference's five full-length known-answer pairs (offline and streamed), and the
stage-tap instantiation against the headline one.  Needs a real MI355X: run with `-m gpu`.

Why these exist: the contract assert (1e-4) has 40x slack over what the path measures, so a
dense 3x3 that dropped its hi.lo / lo.hi products would pass it and keep the faster number; and the parity inputs were
short and synthetic while the reference ships five 31-second real-speech pairs."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_params, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as graft
    graft.build()
    return torch.device("cuda:0")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ---------------------------------------------------------------------------------------------- (a) against float64
@pytest.mark.parametrize("tag", ["dns3", "rand"])
@pytest.mark.parametrize("B,T", [(4, 64), (2, 251)])
def test_forward_is_fp32_class_against_float64(dev, tag, B, T):
    """HIP forward vs the float64 evaluation of the same eval-mode graph (oracle/torch_port.py, dtype=float64), next to
    the fp32 PyTorch-CPU port's own error against it -- ATen fp32 is what the reference runs, so that is what
    "fp32-class" means.  Bounds: rms <= 1.5x the port's rms; max <= 2x the port's max (the maximum over 1.3e5 .. 2.6e5
    elements is an extreme-value statistic: round 3 measured ratios 1.29 and 1.47 at B=4, T=64) and, absolutely,
    <= 3e-6 of the output scale.  For scale: the fp32 chain measures 1.0e-6 / 3.1e-8 (max / rms), a two-plane bf16 split
    with three products 7.8e-6 / 1.5e-6 in the dense phase alone (profiles/r03_ubench_split_bf16.txt)."""
    from gtcrn_micro_amd import Engine
    from oracle.torch_port import TorchPort
    blob = load_params(tag)
    rng = np.random.default_rng(5 if T == 64 else 6)
    spec = (rng.standard_normal((B, 257, T, 2)) * 0.3).astype(np.float32)
    t64 = TorchPort(blob, dtype=torch.float64).forward(torch.from_numpy(spec)).numpy()
    t32 = TorchPort(blob).forward(torch.from_numpy(spec)).numpy()
    ours = Engine(blob, 0).forward_spec(cu(spec)).cpu().numpy().astype(np.float64)
    sc = np.abs(t64).max()
    e_max, e_rms = np.abs(ours - t64).max() / sc, np.sqrt(np.mean((ours - t64) ** 2)) / sc
    p_max, p_rms = np.abs(t32 - t64).max() / sc, np.sqrt(np.mean((t32 - t64) ** 2)) / sc
    print(f"[{tag} B={B} T={T}] HIP vs fp64: max {e_max:.2e} rms {e_rms:.2e}; PyTorch-CPU fp32 vs fp64: max {p_max:.2e} "
          f"rms {p_rms:.2e}; ratios {e_max / p_max:.2f} / {e_rms / p_rms:.2f}")
    assert e_rms <= 1.5 * p_rms, (e_rms, p_rms)
    assert e_max <= 2.0 * p_max, (e_max, p_max)
    assert e_max <= 3e-6, e_max


def test_wave_to_wave_is_fp32_class_against_float64(dev):
    """The headline's own boundary (wave -> wave, the caller loop of infer.py:60-76) against float64: STFT, model and
    iSTFT of the port all in double."""
    from gtcrn_micro_amd import Engine
    from oracle.torch_port import TorchPort
    blob = load_params("dns3")
    rng = np.random.default_rng(9)
    wave = (rng.standard_normal((2, 64000)) * 0.1).astype(np.float32)
    win32 = torch.hann_window(512).pow(0.5)

    def port(dtype):
        x = torch.from_numpy(wave).to(dtype)
        w = win32.to(dtype)
        spec = torch.view_as_real(torch.stft(x, 512, 256, 512, w, return_complex=True))
        out = TorchPort(blob, dtype=dtype).forward(spec)
        return torch.istft(torch.view_as_complex(out.contiguous()), 512, 256, 512, w).numpy()

    t64, t32 = port(torch.float64), port(torch.float32)
    ours = Engine(blob, 0).forward_wave(cu(wave), win32.cuda()).cpu().numpy().astype(np.float64)
    sc = np.abs(t64).max()
    e_max, e_rms = np.abs(ours - t64).max() / sc, np.sqrt(np.mean((ours - t64) ** 2)) / sc
    p_max, p_rms = np.abs(t32 - t64).max() / sc, np.sqrt(np.mean((t32 - t64) ** 2)) / sc
    print(f"[wave->wave] HIP vs fp64: max {e_max:.2e} rms {e_rms:.2e}; PyTorch-CPU fp32 vs fp64: max {p_max:.2e} rms {p_rms:.2e}")
    assert e_rms <= 1.5 * p_rms, (e_rms, p_rms)
    assert e_max <= 2.0 * p_max and e_max <= 3e-6, (e_max, p_max)


# ------------------------------------------------------------------------------- (c) the split itself, on the device
def _bf16_rne(x):
    """fp32 -> bf16 (round to nearest even) -> fp32, in numpy integer arithmetic."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def test_device_split3_is_exact_on_adversarial_values(dev):
    """split3 / join3 as the kernels run them (v_cvt_pk_bf16_f32, fp32 subtractions), on values chosen to break them:
    powers of two and their neighbours (+-1, +-2 ulp), mantissas of all ones, values whose mid / lo planes vanish,
    huge (1e30) and tiny magnitudes down to where the lo plane leaves the normal range, +-0, and random bit patterns.
    Claims: (1) hi == bf16_rne(x), mid == bf16_rne(x - hi), lo == x - hi - mid, bit for bit against a numpy integer
    model; (2) hi + mid + lo == x EXACTLY (in float64) and join3 returns x bit for bit -- for every value whose three
    planes are normal numbers or zero (|x| >= 2^-102: below that the lo plane would be a bf16 subnormal; the model's
    activations are nowhere near); (3) the sign of zero survives."""
    from gtcrn_micro_amd import selftest_split3
    rng = np.random.default_rng(3)
    vals = []
    for k in range(-100, 100, 3):
        b = np.float32(2.0) ** np.float32(k)
        for ulps in (-2, -1, 0, 1, 2):
            vals.append((b.view(np.uint32) + np.uint32(ulps & 0xFFFFFFFF)).astype(np.uint32).view(np.float32))
    vals = [np.float32(v) for v in vals]
    vals += [np.float32(1e30), np.float32(-1e30), np.float32(3.3e38), np.float32(1.0 + 2.0 ** -8), np.float32(1.0 + 2.0 ** -9),
             np.float32(1.0 + 2.0 ** -16), np.float32(1.0 + 2.0 ** -17), np.float32(1.0 - 2.0 ** -24), np.float32(0.0),
             np.float32(-0.0), np.float32(2.0 ** -100)]
    vals += list(np.array([0x3FFFFFFF, 0x3F7FFFFF, 0x3F80FFFF, 0x3F807FFF, 0x3F808000, 0x3F808001, 0x3F817F80,
                           0xBF7FFFFF, 0x477FE000, 0x00800000 + (30 << 23)], np.uint32).view(np.float32))
    u = rng.integers(0, 2 ** 32, size=20000, dtype=np.uint64).astype(np.uint32)
    e = (u >> 23) & 0xFF
    u = u[(e > 40) & (e < 250)]                                             # finite, all planes normal
    x = np.concatenate([np.array(vals, np.float32), u.view(np.float32), (rng.standard_normal(4096) * 3).astype(np.float32)])
    x = x[: (x.size // 4) * 4]
    planes, joined = selftest_split3(x)
    hi = _bf16_rne(x)
    r1 = (x - hi).astype(np.float32)
    mid = _bf16_rne(r1)
    lo_want = _bf16_rne((r1 - mid).astype(np.float32))
    assert np.array_equal(planes[0].view(np.uint32), hi.view(np.uint32)), "hi plane"
    assert np.array_equal(planes[1].view(np.uint32), mid.view(np.uint32)), "mid plane"
    assert np.array_equal(planes[2].view(np.uint32), lo_want.view(np.uint32)), "lo plane"
    s64 = planes[0].astype(np.float64) + planes[1].astype(np.float64) + planes[2].astype(np.float64)
    assert np.array_equal(s64, x.astype(np.float64)), "hi + mid + lo != x"
    assert np.array_equal(joined.view(np.uint32)[x != 0], x.view(np.uint32)[x != 0]), "join3(split3(x)) != x"
    z = x == 0
    assert np.array_equal(np.signbit(planes[0][z]), np.signbit(x[z])) and np.all(joined[z] == 0)
    # at the bottom of the exponent range (|x| ~ 1e-38 * 2^16 and below) the lo / mid planes leave the normal numbers:
    # whatever the hardware does with them (keep, flush), the planes must still add up to x within 2^-125 ABSOLUTE
    tiny = np.array([1e-38 * 65536.0, -1e-38 * 65536.0, 1.1e-38 * 256.0, 2.0 ** -120 * 1.2345678, 1.3e-38, 1e-40, -1e-42, 0.0],
                    np.float32)
    pt, jt = selftest_split3(tiny)
    st = pt[0].astype(np.float64) + pt[1].astype(np.float64) + pt[2].astype(np.float64)
    assert np.isfinite(pt).all() and np.isfinite(jt).all()
    assert np.abs(st - tiny.astype(np.float64)).max() <= 2.0 ** -125, np.abs(st - tiny.astype(np.float64)).max()
    assert np.abs(jt.astype(np.float64) - tiny.astype(np.float64)).max() <= 2.0 ** -125


def test_device_split_product_keeps_fp32_accuracy(dev):
    """split_mm6 -- the ONE helper the dense 3x3, de_convs.3 and this test share -- on a 16x32 by 32x16 product with
    operands split on the device: against the float64 product the error must be that of an fp32 accumulation
    (<= 4e-7 of sum|a||b| per element; measured ~1e-7), which a product set without hi.lo / lo.hi (~1.5e-5 relative
    per term) or a two-plane split (~4e-6) cannot meet; and adversarial operands whose information sits ONLY in the
    mid / lo planes must still multiply exactly."""
    from gtcrn_micro_amd import selftest_split3
    rng = np.random.default_rng(12)
    x = np.zeros(4, np.float32)
    worst = 0.0
    for trial in range(8):
        A = (rng.standard_normal((16, 32)) * (10.0 ** rng.uniform(-2, 2))).astype(np.float32)
        B = (rng.standard_normal((32, 16)) * (10.0 ** rng.uniform(-2, 2))).astype(np.float32)
        _, _, D = selftest_split3(x, A, B)
        want = A.astype(np.float64) @ B.astype(np.float64)
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        worst = max(worst, float((np.abs(D - want) / scale).max()))
    print(f"split_mm6: worst |err| / sum|a||b| = {worst:.2e}")
    assert worst <= 4e-7, worst
    # information only below the hi plane: a = 1 + 2^-12 (mid plane), b = 1 + 2^-20 (lo plane), one live k
    A = np.zeros((16, 32), np.float32)
    B = np.zeros((32, 16), np.float32)
    A[:, 0] = np.float32(1.0 + 2.0 ** -12)
    B[0, :] = np.float32(1.0 + 2.0 ** -20)
    _, _, D = selftest_split3(x, A, B)
    want = (1.0 + 2.0 ** -12) * (1.0 + 2.0 ** -20)
    assert np.abs(D.astype(np.float64) - want).max() <= 2.0 ** -23, np.abs(D.astype(np.float64) - want).max()
    assert np.abs(D.astype(np.float64) - 1.0).min() > 2.0 ** -13        # the mid-plane term is there at all


# ------------------------------------------------------------------- (d) the reference's own known-answer pairs, whole
@pytest.fixture(scope="module")
def examples():
    return np.load(os.path.join(GOLDEN, "examples_full.npz"))


def test_five_reference_example_pairs_offline_full_length(dev, examples):
    """examples/gtcrn_micro/noisy{1..5}.wav -> enh{1..5}.wav at full length (31 s, T = 1 938 frames, 122 chunks of 16
    frames): gtcrn_forward_wave on all five clips, <= 1 LSB of the int16 files (+ float noise: the reference's own
    re-run in the build container is at 1.0004 .. 1.0068 LSB, tests/golden/MANIFEST_examples.json -- its wav writer
    truncated)."""
    from gtcrn_micro_amd import Engine
    eng = Engine(load_params("dns3"), 0)
    win = torch.hann_window(512).pow(0.5).cuda()                     # exactly what infer.py:65 passes
    x = cu(examples["noisy"].astype(np.float32) / 32768.0)
    y = eng.forward_wave(x, win)
    assert tuple(y.shape) == examples["enh"].shape == (5, 495872)
    lsb = np.abs(y.cpu().numpy().astype(np.float64) * 32768.0 - examples["enh"]).max(axis=1)
    print("offline, max deviation from enhN.wav in LSB:", np.round(lsb, 4))
    assert (lsb <= 1.05).all(), lsb
    # one clip at a time (infer.py's own loop, B = 1) gives the same samples bit for bit
    for i in (0, 4):
        assert torch.equal(eng.forward_wave(x[i], win), y[i]), i


def test_five_reference_example_pairs_streamed_frame_by_frame(dev, examples):
    """The same five clips through StreamGTCRNMicro.forward(spec_t, conv_cache, tra_cache, tcn_cache) frame by frame
    (1 938 calls; the loop of gtcrn_micro_stream.py:618-635, five streams side by side), caller-owned caches handed
    back every frame: the iSTFT of the streamed spectrogram is within 1 LSB of the reference's files, and the streamed
    spectrogram equals the offline forward BIT FOR BIT over all 1 938 frames (every ring index wraps many times: the
    deepest TCN ring holds 16 rows).  Then clip 1 alone through the native step in ragged chunks
    (1, 5, 6, 10, 11, 16, 17, 33, ... frames: every tiles-per-wave instantiation and kernel form)."""
    import gtcrn_micro_amd as G
    from gtcrn_micro_amd import Engine
    from gtcrn_micro_amd.models.gtcrn_micro import load_blob_into
    from gtcrn_micro_amd.streaming.gtcrn_micro_stream import StreamGTCRNMicro
    blob = load_params("dns3")
    win = torch.hann_window(512).pow(0.5).cuda()
    x = cu(examples["noisy"].astype(np.float32) / 32768.0)
    spec = G.stft(x, win)                                            # (5,257,1938,2)
    T = spec.shape[2]
    assert T == 1938
    sm = StreamGTCRNMicro().eval()
    load_blob_into(sm, blob)
    sm = sm.to("cuda")
    conv_cache, tra_cache, tcn_cache = sm.init_caches(5, "cuda")
    ys = []
    with torch.no_grad():
        for t in range(T):
            y, conv_cache, tra_cache, tcn_cache = sm(spec[:, :, t:t + 1], conv_cache, tra_cache, tcn_cache)
            ys.append(y)
    streamed = torch.cat(ys, 2)
    eng = Engine(blob, 0)
    offline = eng.forward_spec(spec)
    assert torch.equal(streamed, offline), float((streamed - offline).abs().max())
    wav = G.istft(streamed, win).cpu().numpy().astype(np.float64)
    lsb = np.abs(wav * 32768.0 - examples["enh"]).max(axis=1)
    print("streamed, max deviation from enhN.wav in LSB:", np.round(lsb, 4))
    assert (lsb <= 1.05).all(), lsb
    assert sm.forward_stats["imports"] == 1                          # the caches were handed back: no re-import
    # ragged chunking of clip 1 through the native step
    st = eng.new_state(1)
    sizes, outs, t = [1, 5, 6, 10, 11, 16, 17, 33, 2, 251, 7, 64, 1, 1, 500], [], 0
    k = 0
    while t < T:
        n = min(sizes[k % len(sizes)], T - t)
        outs.append(eng.stream_step(st, spec[:1, :, t:t + n]))
        t += n
        k += 1
    assert torch.equal(torch.cat(outs, 2), offline[:1])


# ------------------------------------------------------------- (e) the stage-tap instantiation vs the headline one
@pytest.mark.parametrize("T", [3, 8, 40])
def test_debug_tap_build_output_equals_headline_instantiation(dev, T):
    """test_forward_every_stage_vs_golden reads its per-stage evidence from k_decoder<DBG = true>, a differently
    scheduled binary than the headline's k_decoder<false>: their OUTPUTS must be the same bits (one, two and three
    tiles per wave)."""
    from gtcrn_micro_amd import Engine
    p = load_params("rand")
    plain, tapped = Engine(p, 0), Engine(p, 0)
    tapped.debug_enable(True)
    rng = np.random.default_rng(T)
    spec = cu((rng.standard_normal((3, 257, T, 2)) * 0.4).astype(np.float32))
    a, b = plain.forward_spec(spec), tapped.forward_spec(spec)
    assert torch.equal(a, b), rel_err(a.cpu().numpy(), b.cpu().numpy())
