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
