"""Pin the CPU oracle against vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden, load_params, rel_err
from oracle import oracle as O

TOL = 2e-5  # oracle vs reference (both fp32 CPU; differences are summation order only)

STAGE_SHAPES = {
    "feat": (3, 17, 257), "erb_bm": (3, 17, 129), "sfe": (3, 17, 129), "en0": (16, 17, 65),
    "en1": (16, 17, 33), "en2": (16, 17, 33), "en3": (16, 17, 33), "en4": (16, 17, 33),
    **{f"gtcn{g}_b{k}": (16, 17, 33) for g in (1, 2) for k in range(4)},
    "de0": (16, 17, 33), "de1": (16, 17, 33), "de2": (16, 17, 33), "de3": (16, 17, 65),
    "de4": (2, 17, 129), "erb_bs": (2, 17, 257),
}


def test_windows():
    g = golden("offline_dns3_T17.npz")
    # the window is caller-owned input (infer.py:65); the helper only has to be close
    assert np.abs(O.window(0) - g["window"]).max() <= 2e-6
    n = np.arange(512)
    assert np.abs(O.window(1) - (0.5 - 0.5 * np.cos(2 * np.pi * n / 512))).max() <= 2e-7


def test_framing_is_bit_exact():
    """frame t = reflect_pad(x,256)[256t:256t+512] * w: indexing must be exact (north_star)."""
    g = golden("offline_dns3_T17.npz")
    x, w = g["wave"], g["window"]
    xp = np.pad(x, 256, mode="reflect")
    want = np.stack([xp[256 * t:256 * t + 512] * w for t in range(1 + len(x) // 256)])
    got = O.frames(x, w)[0]
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("tag", ["dns3", "rand"])
def test_stft_istft(tag):
    g = golden(f"offline_{tag}_T17.npz")
    spec = O.stft(g["wave"], g["window"])
    assert spec.shape == g["spec"].shape
    assert rel_err(spec, g["spec"]) < 2e-6
    wav = O.istft(g["spec_enh"], g["window"])
    assert wav.shape[1] == g["wave_out"].shape[-1] == 4096
    assert rel_err(wav[0], g["wave_out"]) < 2e-6


def test_stft_roundtrip():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 256 * 9)).astype(np.float32)
    w = O.window(0)
    y = O.istft(O.stft(x, w), w)
    assert y.shape == x.shape and np.abs(y - x).max() < 2e-6


@pytest.mark.parametrize("tag", ["dns3", "rand"])
def test_offline_every_stage(tag):
    g = golden(f"offline_{tag}_T17.npz")
    orc = O.Oracle(load_params(tag))
    out = orc.forward(g["spec"])
    for name, shp in STAGE_SHAPES.items():
        got = orc.tap(name, shp)
        want = g[name][0]
        assert want.shape == shp, (name, want.shape)
        assert rel_err(got, want) < TOL, name
    assert rel_err(out, g["spec_enh"]) < TOL


def test_offline_batch():
    g = golden("offline_dns3_B3_T12.npz")
    orc = O.Oracle(load_params("dns3"))
    spec = O.stft(g["wave"], O.window(0))
    assert rel_err(spec, g["spec"]) < 2e-6
    out = orc.forward(g["spec"])
    assert rel_err(out, g["spec_enh"]) < TOL
    assert rel_err(O.istft(out, O.window(0)), g["wave_out"]) < TOL
    assert rel_err(orc.enhance(g["wave"]), g["wave_out"]) < TOL


@pytest.mark.parametrize("tag", ["dns3", "rand"])
def test_streaming_matches_reference_caches(tag):
    g = golden(f"stream_{tag}_T17.npz")
    orc = O.Oracle(load_params(tag))
    st = O.new_states(1)
    spec = g["spec"]
    outs = []
    for i in range(spec.shape[2]):
        outs.append(orc.forward(spec[:, :, i:i + 1], st))
        if i in (0, 1, 16):
            conv, tra, tcn = O.state_views(st[0])
            assert rel_err(conv, g[f"conv_cache_f{i}"][:, 0]) < TOL
            assert rel_err(tra, g[f"tra_cache_f{i}"][:, :, 0]) < TOL
            for gi in range(2):
                for k in range(4):
                    assert rel_err(tcn[gi][k], g[f"tcn_cache_f{i}_g{gi}_b{k}"][0]) < TOL, (i, gi, k)
    ys = np.concatenate(outs, axis=2)
    assert rel_err(ys, g["spec_enh_stream"]) < TOL
    assert rel_err(ys, g["spec_enh_offline"]) < TOL


def test_chunked_equals_offline():
    """Any chunking of T with carried state equals the offline result (streaming == offline)."""
    g = golden("offline_rand_T17.npz")
    orc = O.Oracle(load_params("rand"))
    full = orc.forward(g["spec"])
    for chunks in ([5, 12], [1, 16], [16, 1], [3, 3, 3, 8]):
        st = O.new_states(1)
        t0, outs = 0, []
        for c in chunks:
            outs.append(orc.forward(g["spec"][:, :, t0:t0 + c], st))
            t0 += c
        assert rel_err(np.concatenate(outs, axis=2), full) < 1e-6, chunks


def test_reference_example_pair():
    """examples/noisy1.wav -> enh1.wav (shipped with the reference) to <= ~1 int16 LSB."""
    g = golden("example_noisy1_head.npz")
    orc = O.Oracle(load_params("dns3"))
    x = g["noisy"].astype(np.float32) / 32768.0
    y = orc.enhance(x)[0]
    n = len(g["enh"])
    assert np.abs(y[:n] * 32768.0 - g["enh"]).max() <= 1.01


def test_reference_causality_test():
    """tests/models/test_gtcrn_micro.py of the reference, against frozen inputs/outputs."""
    g = golden("causality_T126.npz")
    orc = O.Oracle(load_params("rand"))
    y1, y2 = orc.enhance(g["x1"])[0], orc.enhance(g["x2"])[0]
    assert np.abs(y1[:16000 - 512] - y2[:16000 - 512]).max() == 0.0
    assert np.abs(y1[16000:] - y2[16000:]).max() > 0
    assert rel_err(y1, g["y1"]) < 5e-5 and rel_err(y2, g["y2"]) < 5e-5


def test_conv_wrappers():
    """tests/streaming/conversion/test_convolution.py of the reference, frozen."""
    g = golden("conv_wrappers.npz")
    # StreamConv2d(1,1,3): offline == frame-by-frame with carried cache
    x, w, b = g["c2d_x"][0], g["c2d_w"], g["c2d_b"]
    y = O.conv2d_causal(x, None, w, b)
    assert np.abs(y - g["c2d_y"][0]).max() < 1e-6
    hist = np.zeros((1, 2, 6), np.float32)
    outs = []
    for i in range(10):
        outs.append(O.conv2d_causal(x[:, i:i + 1], hist, w, b))
        hist = np.concatenate([hist, x[:, i:i + 1]], axis=1)[:, 1:]
    assert np.abs(np.concatenate(outs, axis=1) - g["c2d_y"][0]).max() < 1e-6
    # StreamConvTranspose2d(4,8,(3,1),dil(2,2),pad(0,1))
    x, w, b = g["ct2d_x"][0], g["ct2d_w"], g["ct2d_b"]
    y = O.convT2d_causal(x, None, w, b, dt=2, df=2, pf=1)
    assert np.abs(y - g["ct2d_y"][0]).max() < 1e-6
    assert np.abs(y - g["ct2d_y_stream"][0]).max() < 1e-6
    # convert_to_stream weight contract (convert.py:35-48): W'[o,i,a,b] = W[i,o,kt-1-a,kf-1-b]
    wp = np.flip(np.transpose(w, (1, 0, 2, 3)), axis=(-2, -1))
    assert np.array_equal(wp, g["ct2d_w_stream"])
    # the model's decoder shape: dense 16->16 (3,3), freq pad 1, first T of T+2 frames
    x, w, b = g["ct33_x"], g["ct33_w"], g["ct33_b"]
    for i in range(2):
        y = O.convT2d_causal(x[i], None, w, b, pf=1)
        assert np.abs(y - g["ct33_y"][i][:, :9]).max() < 2e-6


@pytest.mark.parametrize("tag", ["dns3", "rand"])
def test_torch_port_matches_reference(tag):
    """The PyTorch-CPU port used as bench.py's cpu_baseline is pinned to the same goldens."""
    from oracle.torch_port import TorchPort
    g = golden(f"offline_{tag}_T17.npz")
    port = TorchPort(load_params(tag))
    assert rel_err(port.forward(g["spec"]).numpy(), g["spec_enh"]) < 2e-6
    assert rel_err(port.enhance(g["wave"][None], g["window"]).numpy()[0], g["wave_out"]) < 2e-6


# ---- train mode: the PyTorch-CPU port (autograd) against the reference's own train step -------------------
@pytest.mark.parametrize("tag", ["dns3", "rand"])
def test_port_train_step_matches_reference(tag):
    """oracle/torch_port.py in train mode is the checker of the HIP training path on shapes without a fixture:
    pin it (forward, HybridLoss, d loss/d enh, all 248 gradients, running statistics) against the fixtures
    the reference generated (tests/golden/make_golden_train.py)."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from oracle.torch_port import TorchPort, is_trainable
    import json
    g = np.load(os.path.join(GOLDEN, f"trainstep_{tag}_B3_T12.npz"))
    blob = np.fromfile(os.path.join(GOLDEN, f"params_{tag}.f32"), np.float32)
    port = TorchPort(blob, train=True)
    taps = {}
    enh, loss, genh, grads = port.train_step(g["noisy_spec"], g["clean_spec"])
    assert np.abs(enh - g["enh"]).max() / np.abs(g["enh"]).max() < 1e-4
    assert abs(loss - float(g["loss"])) / float(g["loss"]) < 1e-5
    assert np.abs(genh - g["grad_enh"]).max() / np.abs(g["grad_enh"]).max() < 2e-3
    assert np.abs(grads - g["grads"]).max() / np.abs(g["grads"]).max() < 2e-3
    assert np.abs(port.blob() - g["params_after"]).max() < 1e-5 * max(1.0, np.abs(g["params_after"]).max())
    man = json.load(open(os.path.join(GOLDEN, "params_manifest.json")))
    n_train = sum(int(np.prod(s)) for n, s, _ in man["tensors"] if is_trainable(n))
    assert n_train == 19014 and sum(1 for n, _, _ in man["tensors"] if is_trainable(n)) == 248
    port2 = TorchPort(blob, train=True)
    port2.forward(g["noisy_spec"], taps)
    for k in ("en0", "en1", "en2", "en4", "gtcn1", "gtcn2", "de0", "de2", "de3", "de4"):
        ref = g["stage:" + k]
        assert np.abs(taps[k].numpy() - ref).max() / np.abs(ref).max() < 1e-4, k


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    """The C restatement runs clean under ASan + UBSan (CPU build; GPU sanitizers are not available on the pool)."""
    import os
    import shutil
    import subprocess
    from conftest import GOLDEN, ROOT
    cc = shutil.which("gcc")
    if cc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "asan_driver")
    src = [os.path.join(ROOT, "oracle", "asan_driver.c"), os.path.join(ROOT, "oracle", "gtcrn_oracle.c")]
    r = subprocess.run([cc, "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-fopenmp", "-I", os.path.join(ROOT, "oracle"), "-o", exe]
                       + src + ["-lm"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", OMP_NUM_THREADS="2")
    r = subprocess.run([exe, os.path.join(GOLDEN, "params_dns3.f32")], capture_output=True, text=True, timeout=600,
                       env=env)
    assert r.returncode == 0 and "asan_driver ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
