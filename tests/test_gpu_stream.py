"""Streaming parity (config 3 path): frame-by-frame == offline, caches == the reference's."""
import numpy as np
import pytest

from conftest import golden, load_params, rel_err

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


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("tag", ["dns3", "rand"])
def test_native_stream_steps_match_reference(dev, tag):
    from gtcrn_micro_amd import Engine
    g = golden(f"stream_{tag}_T17.npz")
    eng = Engine(load_params(tag), 0)
    spec = cu(g["spec"])
    st = eng.new_state(1)
    outs = []
    for i in range(17):
        outs.append(eng.stream_step(st, spec[:, :, i:i + 1]))
        if i in (0, 1, 16):
            conv = torch.zeros(2, 1, 16, 6, 33, device="cuda")
            tra = torch.zeros(2, 3, 1, 8, 2, device="cuda")
            tcn = [[torch.zeros(1, 16, 2 * d, 33, device="cuda") for d in (1, 2, 4, 8)] for _ in range(2)]
            eng.stream_export(st, conv, tra, tcn)
            assert rel_err(conv.cpu().numpy(), g[f"conv_cache_f{i}"]) < TOL, i
            assert rel_err(tra.cpu().numpy(), g[f"tra_cache_f{i}"]) < TOL, i
            for gi in range(2):
                for k in range(4):
                    assert rel_err(tcn[gi][k].cpu().numpy(), g[f"tcn_cache_f{i}_g{gi}_b{k}"]) < TOL, (i, gi, k)
    ys = torch.cat(outs, dim=2).cpu().numpy()
    assert rel_err(ys, g["spec_enh_stream"]) < TOL
    assert rel_err(ys, g["spec_enh_offline"]) < TOL


def test_stream_module_mirror_reference_call(dev):
    """StreamGTCRNMicro.forward(spec, conv_cache, tra_cache, tcn_cache) with caller-owned caches."""
    import json, os
    from conftest import GOLDEN
    from gtcrn_micro_amd.models.gtcrn_micro import GTCRNMicro
    from gtcrn_micro_amd.streaming.gtcrn_micro_stream import StreamGTCRNMicro
    from gtcrn_micro_amd.streaming.conversion.convert import convert_to_stream
    g = golden("stream_rand_T17.npz")
    p = load_params("rand")
    man = json.load(open(os.path.join(GOLDEN, "params_manifest.json")))
    model = GTCRNMicro().eval()
    model.load_state_dict({n: torch.from_numpy(p[o:o + int(np.prod(s))].reshape(s).copy())
                           for n, s, o in man["tensors"]}, strict=False)
    sm = StreamGTCRNMicro().eval()
    convert_to_stream(sm, model)
    sm = sm.to("cuda")
    conv_cache, tra_cache, tcn_cache = sm.init_caches(1, "cuda")
    spec = cu(g["spec"])
    ys = []
    with torch.no_grad():
        for i in range(17):
            y, conv_cache, tra_cache, tcn_cache = sm(spec[:, :, i:i + 1], conv_cache, tra_cache, tcn_cache)
            ys.append(y)
    assert rel_err(torch.cat(ys, 2).cpu().numpy(), g["spec_enh_stream"]) < TOL
    assert rel_err(conv_cache.cpu().numpy(), g["conv_cache_f16"]) < TOL
    with pytest.raises(AssertionError):
        sm(spec[:, :, :1], conv_cache[:, :, :, :4], tra_cache, tcn_cache)


def test_chunked_equals_offline_many_streams(dev):
    from gtcrn_micro_amd import Engine
    eng = Engine(load_params("rand"), 0)
    rng = np.random.default_rng(3)
    N, T = 64, 50
    spec = cu((rng.standard_normal((N, 257, T, 2)) * 0.3).astype(np.float32))
    full = eng.forward_spec(spec)
    # every kernel instantiation: one / two / three tiles per wave (calls of <= 5, <= 10, more frames), rings in the
    # stream state (one-chunk calls) or copied through LDS (longer calls)
    for chunks in ([1] * 50, [7, 16, 16, 11], [33, 17], [3, 1, 2, 44], [5, 6, 10, 11, 4, 14], [2] * 25, [9, 10, 15, 16]):
        st = eng.new_state(N)
        t0, outs = 0, []
        for c in chunks:
            outs.append(eng.stream_step(st, spec[:, :, t0:t0 + c]))
            t0 += c
        got = torch.cat(outs, 2)
        assert t0 == T
        # every instantiation rounds identically (-ffp-contract=on): chunked == offline BIT FOR BIT
        assert torch.equal(got, full), (chunks, rel_err(got.cpu().numpy(), full.cpu().numpy()))


def test_stream_counts_that_do_not_fill_a_workgroup(dev):
    """Single-frame steps serve four streams per workgroup: stream counts that leave the last workgroup with 1, 2 or 3
    live streams (and a lone stream) still equal the offline result bit for bit, and mixing single-frame steps with
    longer calls keeps the per-stream state consistent between the two kernel forms."""
    from gtcrn_micro_amd import Engine
    eng = Engine(load_params("dns3"), 0)
    rng = np.random.default_rng(11)
    T = 21
    for N in (1, 2, 3, 5, 6, 7):
        spec = cu((rng.standard_normal((N, 257, T, 2)) * 0.3).astype(np.float32))
        full = eng.forward_spec(spec)
        for chunks in ([1] * T, [1, 1, 4, 1, 1, 1, 9, 1, 1, 1]):
            st = eng.new_state(N)
            t0, outs = 0, []
            for c in chunks:
                outs.append(eng.stream_step(st, spec[:, :, t0:t0 + c]))
                t0 += c
            assert t0 == T
            assert torch.equal(torch.cat(outs, 2), full), (N, chunks)


def test_conv_wrappers_like_reference_test(dev):
    """tests/streaming/conversion/test_convolution.py of the reference, run against the HIP wrappers:
    streaming == offline for StreamConv2d(1,1,3) and StreamConvTranspose2d(4,8,(3,1),dil(2,2),pad(0,1)),
    including convert_to_stream's weight permute + flip.  Offline results come from the frozen goldens."""
    import torch.nn as nn
    from gtcrn_micro_amd.streaming.conversion.convert import convert_to_stream
    from gtcrn_micro_amd.streaming.conversion.convolution import StreamConv2d, StreamConvTranspose2d
    g = golden("conv_wrappers.npz")
    # --- StreamConv2d(1, 1, 3)
    conv = nn.Conv2d(1, 1, 3)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(g["c2d_w"])); conv.bias.copy_(torch.from_numpy(g["c2d_b"]))
    sconv = StreamConv2d(1, 1, 3)
    convert_to_stream(stream_model=sconv, model=conv)
    sconv = sconv.cuda()
    x = cu(g["c2d_x"])
    cache = torch.zeros(1, 1, 2, 6, device="cuda")
    outs = []
    for i in range(10):
        o, cache = sconv(x[:, :, i:i + 1], cache)
        outs.append(o)
    assert np.abs(torch.cat(outs, 2).cpu().numpy() - g["c2d_y"]).max() < 1e-6
    o_all, _ = sconv(x, torch.zeros(1, 1, 2, 6, device="cuda"))                # whole clip in one call
    assert np.abs(o_all.cpu().numpy() - g["c2d_y"]).max() < 1e-6
    # --- StreamConvTranspose2d(4, 8, (3,1), stride (1,1), padding (0,1), dilation (2,2))
    kt, dt = 3, 2
    pt = (kt - 1) * dt
    de = nn.ConvTranspose2d(4, 8, (kt, 1), stride=(1, 1), padding=(pt, 1), dilation=(dt, 2), groups=1)
    with torch.no_grad():
        de.weight.copy_(torch.from_numpy(g["ct2d_w"])); de.bias.copy_(torch.from_numpy(g["ct2d_b"]))
    sde = StreamConvTranspose2d(4, 8, (kt, 1), stride=(1, 1), padding=(0, 1), dilation=(dt, 2), groups=1)
    convert_to_stream(sde, de)
    assert np.array_equal(sde.ConvTranspose2d.weight.detach().numpy(), g["ct2d_w_stream"])
    sde = sde.cuda()
    x = cu(g["ct2d_x"])
    cache = torch.zeros(1, 4, pt, 6, device="cuda")
    outs = []
    for i in range(100):
        o, cache = sde(x[:, :, i:i + 1], cache)
        outs.append(o)
    assert np.abs(torch.cat(outs, 2).cpu().numpy() - g["ct2d_y"]).max() < 1e-6
    with pytest.raises(AssertionError):
        StreamConv2d(1, 1, 3, padding=(1, 0))            # time padding must be 0 (convolution.py:94)
    # the C ABI's transposed form on the OFFLINE weight layout equals the same thing
    from gtcrn_micro_amd._lib import stream_conv2d
    y, _ = stream_conv2d(cu(g["ct33_x"]), None, cu(g["ct33_w"]), cu(g["ct33_b"]), 3, 3, pad_f=1, transposed=True)
    assert np.abs(y.cpu().numpy() - g["ct33_y"][:, :, :9]).max() < 2e-6


def test_stream_conv1d_streaming_equals_offline(dev):
    """StreamConv1d (streaming/conversion/convolution.py:10-59): frame-by-frame == nn.Conv1d over the padded clip
    (checker: plain PyTorch fp32 on the CPU), incl. dilation and groups; out_cache = cat([cache, x])[..., 1:]."""
    import torch.nn as nn
    from gtcrn_micro_amd.streaming.conversion.convolution import StreamConv1d
    torch.manual_seed(2)
    for cin, cout, k, d, groups in ((8, 8, 3, 1, 8), (4, 6, 3, 2, 1)):
        ref = nn.Conv1d(cin, cout, k, dilation=d, groups=groups)
        sc = StreamConv1d(cin, cout, k, dilation=d, groups=groups)
        sc.Conv1d.load_state_dict(ref.state_dict())
        sc = sc.cuda()
        x = torch.randn(2, cin, 20)
        H = (k - 1) * d
        want = ref(torch.nn.functional.pad(x, [H, 0])).detach().numpy()
        cache = torch.zeros(2, cin, H, device="cuda")
        outs = []
        for i in range(20):
            o, cache = sc(x[:, :, i:i + 1].cuda(), cache)
            assert cache.shape == (2, cin, H)
            outs.append(o)
        assert np.abs(torch.cat(outs, 2).cpu().numpy() - want).max() < 1e-5
        assert torch.equal(cache.cpu(), x[:, :, 20 - H:])


def test_config3_shape_1024_streams_single_frame_calls(dev):
    """BASELINE configs[2] at its own shape: 1024 concurrent streams, one 16 ms frame per call (the loop of
    gtcrn_micro_stream.py:618-635 with the state left on the device), 24 calls.  A sample of streams is checked against
    the CPU oracle run frame by frame with its own caches; ALL streams against the offline forward of the same frames
    (streaming == offline is the reference's own contract, tests/streaming/...); a batch of 1024 equals the same
    streams run in batches of 64 bit for bit (4 workgroup rounds per launch must not leak between streams)."""
    from gtcrn_micro_amd import Engine
    from oracle import oracle as O
    p = load_params("dns3")
    eng = Engine(p, 0)
    N, T = 1024, 24
    gen = torch.Generator(device="cuda").manual_seed(1024)
    spec = (torch.randn(N, T, 257, 2, device="cuda", generator=gen) * 0.3).permute(0, 2, 1, 3)   # frame-major storage
    spec[5] = 0.0                                        # one silent stream
    spec[6] *= 30.0                                      # one hot stream
    st = eng.new_state(N)
    outs = [eng.stream_step(st, spec[:, :, t:t + 1]) for t in range(T)]
    got = torch.cat(outs, 2)
    assert got.shape == (N, 257, T, 2) and bool(torch.isfinite(got).all())
    full = eng.forward_spec(spec.contiguous())
    err = (got - full).abs().amax(dim=(1, 2, 3)) / full.abs().amax(dim=(1, 2, 3)).clamp_min(1e-20)
    assert torch.equal(got, full), float(err.max())          # streamed == offline bit for bit, all 1024 streams
    # oracle: streams 0, 5 (silence), 6 (hot), 511, 1023, frame by frame with the reference cache layout
    orc = O.Oracle(p)
    for s in (0, 5, 6, 511, 1023):
        x = spec[s:s + 1].contiguous().cpu().numpy()
        states = O.new_states(1)
        ref = np.concatenate([orc.forward(x[:, :, t:t + 1], states) for t in range(T)], axis=2)
        if s == 5:
            assert np.abs(got[s].cpu().numpy()).max() < 1e-6 and np.abs(ref).max() < 1e-6
        else:
            assert rel_err(got[s:s + 1].cpu().numpy(), ref) < TOL, s
    # batch invariance: the same streams in 16 batches of 64
    parts = []
    for lo in range(0, N, 64):
        st64 = eng.new_state(64)
        parts.append(torch.cat([eng.stream_step(st64, spec[lo:lo + 64, :, t:t + 1]) for t in range(T)], 2))
    assert torch.equal(torch.cat(parts, 0), got)
    # the exported caches of stream 1023 equal the oracle's after the same 24 frames
    conv = torch.zeros(2, N, 16, 6, 33, device="cuda")
    tra = torch.zeros(2, 3, N, 8, 2, device="cuda")
    tcn = [[torch.zeros(N, 16, 2 * d, 33, device="cuda") for d in (1, 2, 4, 8)] for _ in range(2)]
    eng.stream_export(st, conv, tra, tcn)
    oc, ot, otcn = O.state_views(states[0])
    assert rel_err(conv[:, 1023].cpu().numpy(), oc) < TOL
    assert rel_err(tra[:, :, 1023].cpu().numpy(), ot) < TOL
    for g in range(2):
        for k in range(4):
            assert rel_err(tcn[g][k][1023].cpu().numpy(), otcn[g][k]) < TOL, (g, k)
