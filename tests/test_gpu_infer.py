"""Bulk offline driver (counterpart of infer.py:26-119) on the shipped example head."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_enhance_folder_matches_reference_example(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as graft
    graft.build()
    from scipy.io import wavfile
    from gtcrn_micro_amd.infer import enhance_folder
    g = golden("example_noisy1_head.npz")
    noisy_dir, clean_dir, enh_dir = tmp_path / "noisy", tmp_path / "clean", tmp_path / "enh"
    noisy_dir.mkdir(); clean_dir.mkdir()
    # two clips of equal length (batched together) and one shorter; clean refs of different lengths
    wavfile.write(noisy_dir / "a_fileid_1.wav", 16000, g["noisy"])
    wavfile.write(noisy_dir / "b_fileid_2.wav", 16000, g["noisy"])
    wavfile.write(noisy_dir / "c_fileid_3.wav", 16000, g["noisy"][:20000])
    wavfile.write(clean_dir / "clean_fileid_1.wav", 16000, np.zeros(32100, np.int16))   # pad
    wavfile.write(clean_dir / "clean_fileid_2.wav", 16000, np.zeros(30000, np.int16))   # crop
    wavfile.write(clean_dir / "clean_fileid_3.wav", 16000, np.zeros(20000, np.int16))
    inf, ref = enhance_folder(str(noisy_dir), str(clean_dir), str(enh_dir),
                              os.path.join(GOLDEN, "params_dns3.f32"), device=0)
    assert [u for u, _ in inf] == ["a_fileid_1", "b_fileid_2", "c_fileid_3"]
    n = len(g["enh"])
    fs, ya = wavfile.read(enh_dir / "a_fileid_1_enh.wav")
    assert fs == 16000 and ya.dtype == np.int16 and len(ya) == 32100
    assert np.abs(ya[:n].astype(np.int32) - g["enh"]).max() <= 2          # reference wav +- rounding
    assert np.all(ya[32000:] == 0)                                         # zero padded to the clean length
    _, yb = wavfile.read(enh_dir / "b_fileid_2_enh.wav")
    assert len(yb) == 30000 and np.array_equal(yb, ya[:30000])
    _, yc = wavfile.read(enh_dir / "c_fileid_3_enh.wav")
    assert len(yc) == 20000
    lines = open(enh_dir / "inf.scp").read().splitlines()
    assert lines[0] == f"a_fileid_1 {enh_dir / 'a_fileid_1_enh.wav'}"
    assert open(enh_dir / "ref.scp").read().splitlines()[2] == f"c_fileid_3 {clean_dir / 'clean_fileid_3.wav'}"
    # missing clean reference -> the reference's FileNotFoundError
    os.remove(clean_dir / "clean_fileid_3.wav")
    with pytest.raises(FileNotFoundError):
        enhance_folder(str(noisy_dir), str(clean_dir), str(enh_dir), os.path.join(GOLDEN, "params_dns3.f32"))


@pytest.mark.gpu
def test_forward_wave_is_graph_capturable():
    """gtcrn_model_reserve pre-sizes the workspace so that a whole wave -> wave call (six kernels, no allocation,
    no synchronisation) can be captured into a HIP graph and replayed on new input."""
    import numpy as np
    import torch
    from conftest import load_params
    from gtcrn_micro_amd import Engine
    eng = Engine(load_params("dns3"), 0)
    B, L = 4, 256 * 40
    win = torch.hann_window(512, device="cuda").pow(0.5)
    wave = torch.randn(B, L, device="cuda") * 0.1
    out = torch.empty(B, L, device="cuda")
    eng.reserve(B, 1 + L // 256)
    ref = eng.forward_wave(wave, win).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        eng.forward_wave(wave, win, out=out)          # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            eng.forward_wave(wave, win, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    wave.copy_(torch.randn(B, L, device="cuda") * 0.1)   # same buffers, new content
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eng.forward_wave(wave, win))


@pytest.mark.gpu
def test_variable_length_batch_equals_single_clips_bit_for_bit(tmp_path):
    """infer.py:48-107 takes the clips of a folder one by one whatever their lengths; here a mixed folder is packed into
    full batches (gtcrn_forward_wave_var, per-clip lengths inside one launch sequence).  Every clip must equal its own
    single-clip result BIT FOR BIT, for lengths on and off the 256-sample hop, around the 16-frame chunk and the
    7-block iSTFT group boundaries, down to the minimum of 257 samples; the folder driver writes the same wavs."""
    import warnings
    from scipy.io import wavfile
    from conftest import load_params
    from gtcrn_micro_amd import Engine, GtcrnError
    from gtcrn_micro_amd.infer import enhance_folder
    from oracle import oracle as O
    p = load_params("dns3")
    eng = Engine(p, 0)
    win = torch.hann_window(512).pow(0.5).cuda()      # computed on the host like infer.py:65 (and enhance_folder)
    lens = [257, 511, 512, 256 * 7, 256 * 7 + 255, 256 * 8, 4095, 4096, 4097, 256 * 33 + 17, 256 * 49, 16000]
    rng = np.random.default_rng(8)
    clips = [(rng.standard_normal(L) * 0.1).astype(np.float32) for L in lens]
    Lmax = max(lens)
    host = np.full((len(lens), Lmax), np.nan, np.float32)      # the padding must never be read: poison it
    for j, c in enumerate(clips):
        host[j, :len(c)] = c
    out = torch.full((len(lens), 256 * (Lmax // 256)), -7.0, device="cuda")
    y = eng.forward_wave_var(torch.from_numpy(host).cuda(), lens, win, out=out)
    # (B = 12 runs in time spans over the lengths' prefix table; one workgroup per utterance gives the same bits)
    eng.var_spans_enable(False)
    y0 = eng.forward_wave_var(torch.from_numpy(host).cuda(), lens, win)
    assert all(torch.equal(y0[j, :256 * (L // 256)], y[j, :256 * (L // 256)]) for j, L in enumerate(lens))
    eng.var_spans_enable(True)
    orc = O.Oracle(p)
    for j, (c, L) in enumerate(zip(clips, lens)):
        single = eng.forward_wave(torch.from_numpy(c).cuda(), win)
        n = 256 * (L // 256)
        assert single.shape[0] == n
        assert torch.equal(y[j, :n], single), (j, L)
        assert bool((y[j, n:] == -7.0).all())                  # the rest of the row is left untouched
        if L in (257, 4097, 16000):
            ref = orc.enhance(c[None])[0]
            assert np.abs(single.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-9) < 1e-4
    with pytest.raises(GtcrnError):
        eng.forward_wave_var(torch.from_numpy(host).cuda(), [256] + lens[1:], win)       # too short to reflect-pad
    with pytest.raises(GtcrnError):
        eng.forward_wave_var(torch.from_numpy(host).cuda(), lens[:-1] + [Lmax + 1], win)
    with pytest.raises(GtcrnError):
        eng.forward_wave(torch.from_numpy(host).cuda(), win, out=torch.empty(3, 5, device="cuda"))   # wrong-sized out
    # the folder driver: 8 clips of 8 different lengths (+ one too short to process) in batches of 4
    noisy_dir, clean_dir, enh_dir = tmp_path / "noisy", tmp_path / "clean", tmp_path / "enh"
    noisy_dir.mkdir(); clean_dir.mkdir()
    pcm = [np.clip(np.rint(c * 32768), -32768, 32767).astype(np.int16) for c in clips[3:11]]
    for k, x in enumerate(pcm):
        wavfile.write(noisy_dir / f"n_fileid_{k}.wav", 16000, x)
        wavfile.write(clean_dir / f"clean_fileid_{k}.wav", 16000, np.zeros(len(x), np.int16))
    wavfile.write(noisy_dir / "n_fileid_99.wav", 16000, np.zeros(100, np.int16))
    wavfile.write(clean_dir / "clean_fileid_99.wav", 16000, np.zeros(100, np.int16))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        inf, _ = enhance_folder(str(noisy_dir), str(clean_dir), str(enh_dir), os.path.join(GOLDEN, "params_dns3.f32"),
                                device=0, max_batch=4)
    assert any("skipped" in str(x.message) for x in w)
    assert [u for u, _ in inf] == [f"n_fileid_{k}" for k in range(8)]
    for k, x in enumerate(pcm):
        single = eng.forward_wave(torch.from_numpy(x.astype(np.float32) / 32768.0).cuda(), win).cpu().numpy()
        want = np.zeros(len(x), np.float32)
        want[:len(single)] = single
        want = np.clip(np.rint(want * 32768.0), -32768, 32767).astype(np.int16)
        _, got = wavfile.read(enh_dir / f"n_fileid_{k}_enh.wav")
        assert np.array_equal(got, want), k


@pytest.mark.gpu
@pytest.mark.parametrize("B,Lmax,seed", [(3, 40000, 1), (40, 48000, 2), (255, 9000, 3), (300, 6000, 4), (1025, 2000, 5)])
def test_variable_length_batches_in_time_spans_equal_one_workgroup_per_utterance(B, Lmax, seed):
    """Variable-length batches that are not whole rounds of 256 share the frames that exist between 256 k workgroups
    (prefix table of the lengths, span_begin in kernels.hip); more than 1024 utterances fall back to one workgroup per
    utterance.  Either way every sample equals the unshared launch bit for bit, including utterances of the minimum
    length and shares that cross several short utterances."""
    from conftest import load_params
    from gtcrn_micro_amd import Engine
    eng = Engine(load_params("dns3"), 0)
    win = torch.hann_window(512).pow(0.5).cuda()
    rng = np.random.default_rng(seed)
    lens = rng.integers(257, Lmax + 1, B).tolist()
    lens[0] = Lmax
    if B > 2:
        lens[1] = 257                                   # one frame pair: shorter than any share's warm-up
        lens[B // 2] = min(Lmax, 256 * 16 + 1)
    wave = torch.randn(B, Lmax, device="cuda", generator=torch.Generator("cuda").manual_seed(seed)) * 0.1
    y = eng.forward_wave_var(wave, lens, win)
    eng.var_spans_enable(False)
    y0 = eng.forward_wave_var(wave, lens, win)
    eng.var_spans_enable(True)
    valid = torch.arange(y.shape[1], device="cuda")[None] < torch.tensor([256 * (L // 256) for L in lens], device="cuda")[:, None]
    assert torch.equal(torch.where(valid, y, 0), torch.where(valid, y0, 0))      # (the rest of a row is unspecified)
    j = B // 2
    single = eng.forward_wave(wave[j, :lens[j]].contiguous(), win)
    assert torch.equal(y[j, :single.shape[0]], single)


@pytest.mark.gpu
def test_pipelined_folder_driver_writes_the_serial_drivers_bytes(tmp_path):
    """The folder driver as a pipeline (reader thread -> pinned staging -> H2D || kernels || D2H on three HIP streams ->
    writer thread) against its serial form (one batch after the other, pageable copies): 40 clips of 8 lengths in batches
    of 6 over 3 staging slots -- more batches than slots, so every slot is reused -- must give byte-identical wav files
    and the same scp lists; the statistics object is filled; a reader failure (a file that disappears) is raised, not hung."""
    from scipy.io import wavfile
    from gtcrn_micro_amd.infer import enhance_folder
    noisy_dir, clean_dir = tmp_path / "noisy", tmp_path / "clean"
    noisy_dir.mkdir(); clean_dir.mkdir()
    rng = np.random.default_rng(11)
    lens = [3000, 4096, 4097, 9000, 9000, 12345, 16000, 20000]
    for k in range(40):
        L = lens[k % len(lens)] + (k // len(lens)) * 7
        x = (rng.standard_normal(L) * 3000).astype(np.int16)
        if k == 3:
            x[:4] = [32767, -32768, 32767, -32768]                  # full scale: the output clips
        # two files are NOT mono 16-bit PCM (32-bit PCM, float): their batches take the generic reader + float32 staging
        wavfile.write(noisy_dir / f"n_fileid_{k}.wav", 16000,
                      x.astype(np.int32) << 16 if k == 17 else (x.astype(np.float32) / 32768.0 if k == 30 else x))
        wavfile.write(clean_dir / f"clean_fileid_{k}.wav", 16000, np.zeros(L + (k % 3) * 100 - 100, np.int16))
    ck = os.path.join(GOLDEN, "params_dns3.f32")
    st_p, st_s = {}, {}
    inf_p, ref_p = enhance_folder(str(noisy_dir), str(clean_dir), str(tmp_path / "enh_p"), ck, max_batch=6, stats=st_p)
    inf_s, ref_s = enhance_folder(str(noisy_dir), str(clean_dir), str(tmp_path / "enh_s"), ck, max_batch=6, pipeline=False,
                                  stats=st_s)
    assert [u for u, _ in inf_p] == [u for u, _ in inf_s] == sorted(f"n_fileid_{k}" for k in range(40))
    assert [r for _, r in ref_p] == [r for _, r in ref_s]
    for u, _ in inf_p:
        a = open(tmp_path / "enh_p" / f"{u}_enh.wav", "rb").read()
        b = open(tmp_path / "enh_s" / f"{u}_enh.wav", "rb").read()
        assert a == b and len(a) > 44, u
    assert st_p["clips"] == st_s["clips"] == 40 and st_p["batches"] == 7 and st_p["frames"] == st_s["frames"]
    assert st_p["pipeline"] is True and 0.0 < st_p["gpu_busy_frac"] <= 1.0 and st_s["gpu_busy_frac"] is None
    # the pipeline moves the samples as 16-bit PCM (converted on the device): half the serial form's bytes out, and in for
    # every batch of plain 16-bit files (the two batches holding the 32-bit and the float file are staged as float32)
    assert 2 * st_p["d2h_bytes"] == st_s["d2h_bytes"] and st_s["h2d_bytes"] / 2 < st_p["h2d_bytes"] < st_s["h2d_bytes"]
    # a failing stage surfaces as an exception of the call (and the threads end): a clip that vanishes after pass 1
    import gtcrn_micro_amd.infer as I
    real = I.read_wav_f32
    calls = {"n": 0}

    def flaky(path):
        calls["n"] += 1
        if calls["n"] == 9:
            raise OSError("simulated read error")
        return real(path)
    I.read_wav_f32 = flaky
    try:
        with pytest.raises(OSError, match="simulated read error"):
            enhance_folder(str(noisy_dir), str(clean_dir), str(tmp_path / "enh_x"), ck, max_batch=6)
    finally:
        I.read_wav_f32 = real
    import threading
    assert not [t for t in threading.enumerate() if t.name.startswith("gtcrn-folder-")]


@pytest.mark.gpu
def test_pcm16_boundary_conversions_are_soundfiles_and_wavfile_writes():
    """gtcrn_pcm16_to_f32 / gtcrn_f32_to_pcm16, the 16-bit PCM boundary of the served pipeline and the folder driver: in,
    every int16 value / 32768 exactly (what soundfile.read hands infer.py:54); out, clip(rint(y * 32768)) with round half
    to even -- np.rint's and write_wav_pcm16's bytes --, including ties (k + 0.5), both clipping ends, -0.0 and a size that
    is several grid strides; int16 -> float -> int16 is the identity; misaligned or odd-sized arguments are refused."""
    import torch
    import gtcrn_micro_amd as G
    allv = torch.arange(-32768, 32768, dtype=torch.int32).to(torch.int16).cuda()
    x = G.pcm16_to_f32(allv)
    assert torch.equal(x.cpu(), allv.cpu().to(torch.float32) / 32768.0)
    assert torch.equal(G.f32_to_pcm16(x), allv)
    rng = np.random.default_rng(5)
    y = np.concatenate([rng.standard_normal(8 * 300001).astype(np.float32) * 0.5,
                        ((np.arange(-40, 40, dtype=np.float32) + 0.5) / 32768.0),             # ties: half to even
                        np.array([1.0, -1.0, 0.99999, -1.00002, 3.5, -7.0, -0.0, 1e-9], np.float32)])
    want = np.clip(np.rint(y * np.float32(32768.0)), -32768, 32767).astype(np.int16)
    got = G.f32_to_pcm16(torch.from_numpy(y).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    with pytest.raises(G.GtcrnError):
        G.f32_to_pcm16(torch.zeros(12, device="cuda"))                       # not a multiple of 8
    with pytest.raises(G.GtcrnError):
        G.pcm16_to_f32(torch.zeros(24, dtype=torch.int16, device="cuda")[4:20])  # 8-byte aligned
    with pytest.raises(G.GtcrnError):
        G.pcm16_to_f32(torch.zeros(16, dtype=torch.int16))                   # host tensor
