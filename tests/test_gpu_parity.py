"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden
vectors.  Needs a real MI355X: run with `-m gpu`.  Tolerance (north_star): 1e-4 relative fp32,
bit-exact for framing/indexing."""
import numpy as np
import pytest

from conftest import REG, REG_STAGE, check_parity, golden, load_params, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as graft
    graft.build()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def engines(dev):
    from gtcrn_micro_amd import Engine
    return {tag: Engine(load_params(tag), 0) for tag in ("dns3", "rand")}


@pytest.fixture(scope="module")
def oracles():
    from oracle import oracle as O
    return {tag: O.Oracle(load_params(tag)) for tag in ("dns3", "rand")}


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_native_library_is_loaded_and_mfma_lane_map(dev):
    from gtcrn_micro_amd import _lib, selftest_mfma
    assert _lib.lib().gtcrn_abi_version() == 1
    selftest_mfma(0)


def test_framing_bit_exact(dev):
    import gtcrn_micro_amd as G
    from oracle import oracle as O
    g = golden("offline_dns3_T17.npz")
    for L in (4096, 257, 1000, 256 * 7 + 255):
        rng = np.random.default_rng(L)
        x = rng.standard_normal((3, L)).astype(np.float32)
        got = G.stft_frames(cu(x), cu(g["window"])).cpu().numpy()
        assert np.array_equal(got, O.frames(x, g["window"])), L


@pytest.mark.parametrize("tag", ["dns3", "rand"])
def test_stft_istft_golden(dev, tag):
    import gtcrn_micro_amd as G
    g = golden(f"offline_{tag}_T17.npz")
    win = cu(g["window"])
    spec = G.stft(cu(g["wave"]), win)
    assert tuple(spec.shape) == (257, 17, 2)
    assert rel_err(spec.cpu().numpy(), g["spec"][0]) < 5e-6
    wav = G.istft(cu(g["spec_enh"]), win)
    assert rel_err(wav.cpu().numpy()[0], g["wave_out"]) < 5e-6


def test_stft_roundtrip_and_layouts(dev):
    import gtcrn_micro_amd as G
    rng = np.random.default_rng(1)
    x = rng.standard_normal((5, 256 * 40 + 17)).astype(np.float32)
    win = cu(G.make_window(0))
    spec = G.stft(cu(x), win)
    y = G.istft(spec, win).cpu().numpy()
    assert y.shape == (5, 256 * 40)
    assert np.abs(y - x[:, :256 * 40]).max() < 5e-6
    # frame-major (B,T,257,2) storage viewed as (B,257,T,2): strides are honoured
    T = spec.shape[2]
    fm = torch.empty((5, T, 257, 2), device="cuda").permute(0, 2, 1, 3)
    G.stft(cu(x), win, out=fm)
    assert torch.equal(fm, spec)
    assert torch.equal(G.istft(fm, win), G.istft(spec, win))


@pytest.mark.parametrize("tag", ["dns3", "rand"])
def test_forward_every_stage_vs_golden(dev, engines, tag):
    g = golden(f"offline_{tag}_T17.npz")
    eng = engines[tag]
    eng.debug_enable(True)
    out = eng.forward_spec(cu(g["spec"])).cpu().numpy()
    errs = {}
    for name in ("en0", "en1", "en2", "en3", "en4", "gtcn1", "gtcn2", "de0", "de1", "de2", "de3", "de4"):
        want = g[{"gtcn1": "gtcn1_b3", "gtcn2": "gtcn2_b3"}.get(name, name)][0]
        errs[name] = rel_err(eng.tap(name, 0, 17), want)
    eng.debug_enable(False)
    errs["out"] = rel_err(out, g["spec_enh"])
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, errs
    bad = {k: v for k, v in errs.items() if not v < (REG if k == "out" else REG_STAGE)}     # regression guard
    assert not bad, errs


def test_forward_batch_and_wave(dev, engines):
    import gtcrn_micro_amd as G
    g = golden("offline_dns3_B3_T12.npz")
    eng = engines["dns3"]
    out = eng.forward_spec(cu(g["spec"]))
    check_parity(out.cpu().numpy(), g["spec_enh"], "B=3 spec")
    win = cu(G.make_window(0))
    y = eng.forward_wave(cu(g["wave"]), win).cpu().numpy()
    check_parity(y, g["wave_out"], "B=3 wave")
    # non-contiguous (frame-major) input and output views
    fm_in = cu(np.transpose(g["spec"], (0, 2, 1, 3)).copy()).permute(0, 2, 1, 3)
    fm_out = torch.empty((3, 12, 257, 2), device="cuda").permute(0, 2, 1, 3)
    eng.forward_spec(fm_in, out=fm_out)
    assert torch.equal(fm_out, out)


@pytest.mark.parametrize("T", [1, 2, 5, 6, 10, 11, 15, 16, 17, 33, 251])   # 5/6 and 10/11: tiles-per-wave switch
def test_forward_lengths_vs_oracle(dev, engines, oracles, T):
    rng = np.random.default_rng(T)
    spec = (rng.standard_normal((2, 257, T, 2)) * 0.5).astype(np.float32)
    got = engines["rand"].forward_spec(cu(spec)).cpu().numpy()
    want = oracles["rand"].forward(spec)
    check_parity(got, want, f"T={T}")


def test_reference_example_pair(dev, engines):
    import gtcrn_micro_amd as G
    g = golden("example_noisy1_head.npz")
    x = g["noisy"].astype(np.float32) / 32768.0
    win = torch.hann_window(512).pow(0.5).cuda()      # exactly what infer.py:65 passes
    y = engines["dns3"].forward_wave(cu(x), win).cpu().numpy()
    n = len(g["enh"])
    assert np.abs(y[:n] * 32768.0 - g["enh"]).max() <= 1.05


def test_reference_causality_test(dev, engines):
    """tests/models/test_gtcrn_micro.py of the reference: shared prefix -> bit-identical output."""
    import gtcrn_micro_amd as G
    g = golden("causality_T126.npz")
    win = cu(G.make_window(0))
    eng = engines["rand"]
    y1 = eng.forward_wave(cu(g["x1"]), win).cpu().numpy()[0]
    y2 = eng.forward_wave(cu(g["x2"]), win).cpu().numpy()[0]
    assert np.abs(y1[:16000 - 512] - y2[:16000 - 512]).max() == 0.0
    assert np.abs(y1[16000:] - y2[16000:]).max() > 0
    check_parity(y1, g["y1"], "causality y1")
    check_parity(y2, g["y2"], "causality y2")


def test_module_mirror_forward(dev):
    from gtcrn_micro_amd.models.gtcrn_micro import GTCRNMicro
    import json, os
    from conftest import GOLDEN
    g = golden("offline_dns3_T17.npz")
    p = load_params("dns3")
    man = json.load(open(os.path.join(GOLDEN, "params_manifest.json")))
    m = GTCRNMicro().eval()
    sd = {name: torch.from_numpy(p[off:off + int(np.prod(shape))].reshape(shape).copy())
          for name, shape, off in man["tensors"]}
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda")
    with torch.inference_mode():
        out = m(cu(g["spec"]))
    check_parity(out.cpu().numpy(), g["spec_enh"], "module forward")
    # infer.py usage: stft(...)[None] -> model -> [0] -> view_as_complex -> istft
    import gtcrn_micro_amd as G
    win = torch.hann_window(512).pow(0.5).cuda()
    spec = G.stft(cu(g["wave"]), win)
    y = m(spec[None])[0]
    wav = G.istft(y, win)
    check_parity(wav.cpu().numpy(), g["wave_out"], "module wave")


def test_full_size_batch_invariance(dev, engines):
    """BASELINE config 2 size (B=256, 4 s clips): every utterance equals its own B=1 result bit
    for bit (no cross-utterance dependence), and streaming the clip in chunks equals offline."""
    import gtcrn_micro_amd as G
    rng = np.random.default_rng(7)
    B, L = 256, 64000
    x = (rng.standard_normal((B, L)) * 0.1).astype(np.float32)
    win = cu(G.make_window(0))
    eng = engines["dns3"]
    xg = cu(x)
    y = eng.forward_wave(xg, win)
    assert tuple(y.shape) == (B, 64000)
    for b in (0, 17, 255):
        y1 = eng.forward_wave(xg[b:b + 1].clone(), win)
        assert torch.equal(y1[0], y[b]), b
    assert bool(torch.isfinite(y).all())


def test_no_dependence_on_stale_memory(dev, engines, oracles):
    """Short, partial chunks after a poisoned large run: results must not depend on what earlier
    launches left in the workspace or in LDS (0 * NaN hazards on out-of-range lanes)."""
    eng = engines["rand"]
    big = torch.full((64, 257, 40, 2), float("nan"), device="cuda")
    eng.forward_spec(big)                                   # poisons workspace and LDS with NaN
    rng = np.random.default_rng(5)
    for T in (1, 3, 12, 15, 21):
        spec = (rng.standard_normal((3, 257, T, 2)) * 0.5).astype(np.float32)
        got = eng.forward_spec(cu(spec)).cpu().numpy()
        assert np.isfinite(got).all(), T
        check_parity(got, oracles["rand"].forward(spec), f"T={T} after poison")


def test_extreme_inputs_stay_finite_and_match_oracle(dev, engines, oracles):
    """Digital silence (the 1e-12 inside the magnitude, zero energies in the TRA gates) and full-scale input."""
    for amp in (0.0, 30.0):
        rng = np.random.default_rng(3)
        spec = (rng.standard_normal((2, 257, 19, 2)) * amp).astype(np.float32)
        got = engines["dns3"].forward_spec(cu(spec)).cpu().numpy()
        want = oracles["dns3"].forward(spec)
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= TOL * max(np.abs(want).max(), 1e-30) + 1e-30


@pytest.mark.parametrize("tag", ["dns3", "rand"])
def test_parity_per_band_and_elementwise(engines, oracles, tag):
    """The headline tolerance is a max-norm ratio (max|a-b| / max|b| <= 1e-4), which cannot see a large relative error
    in a quiet band.  Here the same forward is judged band by band and element by element on a speech-like input whose
    spectrum falls 40 dB from the low to the high bins (plus a louder, flat noise floor in a second clip):
      * relative L2 error per band of 16 bins, per frame block:  <= 1e-4 in every band;
      * element-wise: |a-b| <= 1e-4 |b| + 1e-5 * (rms of b's own bin) for >= 99.5 % of the elements (measured 99.77 % / 99.99 %; the absolute
        term covers outputs that are small by cancellation in the complex mask product re*mr - im*mi: both sides
        are fp32 with different summation orders; with a 1e-6 floor 98.9 % / 99.97 % of the elements pass)."""
    rng = np.random.default_rng(77)
    T = 40
    tilt = 10.0 ** (-2.0 * np.arange(257) / 256.0)            # -40 dB across the band
    spec = np.stack([rng.standard_normal((257, T, 2)) * tilt[:, None, None] * 3.0,
                     rng.standard_normal((257, T, 2)) * tilt[:, None, None] * 0.02 + rng.standard_normal((257, T, 2)) * 0.002])
    spec = spec.astype(np.float32)
    got = engines[tag].forward_spec(cu(spec)).cpu().numpy().astype(np.float64)
    ref = oracles[tag].forward(spec).astype(np.float64)
    worst_band = 0.0
    for b in range(2):
        for lo in range(0, 257, 16):
            hi = min(lo + 16, 257)
            for t0 in range(0, T, 10):
                a, r = got[b, lo:hi, t0:t0 + 10], ref[b, lo:hi, t0:t0 + 10]
                worst_band = max(worst_band, float(np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-30)))
    bin_rms = np.sqrt((ref ** 2).mean(axis=(2, 3), keepdims=True))
    ok = np.abs(got - ref) <= 1e-4 * np.abs(ref) + 1e-5 * bin_rms
    frac = float(ok.mean())
    frac6 = float((np.abs(got - ref) <= 1e-4 * np.abs(ref) + 1e-6 * bin_rms).mean())
    print(f"per-band worst relative L2 [{tag}]: {worst_band:.2e}; elements within 1e-4 relative (+1e-5 / +1e-6 of the "
          f"bin rms): {100 * frac:.3f} % / {100 * frac6:.3f} %")
    assert worst_band < TOL, worst_band
    assert frac >= 0.995, frac
    check_parity(got, ref, "tilted spectrum")


@pytest.mark.parametrize("B,T", [(1, 300), (2, 251), (3, 17), (5, 90), (31, 64), (257, 40), (300, 33), (513, 20)])
def test_time_spans_equal_one_workgroup_per_utterance_bit_for_bit(dev, engines, B, T):
    """Offline calls whose batch does not fill the 256 CUs cut the (utterance, frame) axis into equal shares per workgroup
    (kernels.hip wg_spans): a share that starts inside an utterance is warmed up from zero history over the kernel's
    receptive field (12 frames for the encoder / decoder blocks, 30 for a GTCN stack) and may run on into the next
    utterance.  Finite receptive field => EXACT: every utterance must equal, bit for bit, its result inside a batch of
    256 copies-plus-others, which runs one workgroup per utterance with no warm-up at all.  B = 257 / 300 / 513: shares
    that cross utterance boundaries (two segments per workgroup); B = 1 .. 31: many shares per utterance."""
    rng = np.random.default_rng(B * 1000 + T)
    spec = cu((rng.standard_normal((B, 257, T, 2)) * 0.4).astype(np.float32))
    eng = engines["rand"]
    got = eng.forward_spec(spec)
    # reference: the same utterances in batches of exactly 256 (nwg == B: no spans), padded with other utterances
    pick = sorted(set([0, B // 2, B - 1] + list(rng.integers(0, B, size=5))))
    pad = cu((rng.standard_normal((256 - len(pick), 257, T, 2)) * 0.4).astype(np.float32))
    ref = eng.forward_spec(torch.cat([spec[pick], pad], 0))[: len(pick)]
    assert torch.equal(got[pick], ref), float((got[pick] - ref).abs().max())
    assert bool(torch.isfinite(got).all())
