"""Streaming parity (config 3 path): frame-by-frame == offline, caches == the reference's."""
import numpy as np
import pytest

from conftest import check_parity, golden, load_params, rel_err

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
            check_parity(conv.cpu().numpy(), g[f"conv_cache_f{i}"], str(i))
            check_parity(tra.cpu().numpy(), g[f"tra_cache_f{i}"], str(i))
            for gi in range(2):
                for k in range(4):
                    check_parity(tcn[gi][k].cpu().numpy(), g[f"tcn_cache_f{i}_g{gi}_b{k}"], str((i, gi, k)))
    ys = torch.cat(outs, dim=2).cpu().numpy()
    check_parity(ys, g["spec_enh_stream"])
    check_parity(ys, g["spec_enh_offline"])


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
    check_parity(torch.cat(ys, 2).cpu().numpy(), g["spec_enh_stream"])
    check_parity(conv_cache.cpu().numpy(), g["conv_cache_f16"])
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
            check_parity(got[s:s + 1].cpu().numpy(), ref, str(s))
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
    check_parity(conv[:, 1023].cpu().numpy(), oc)
    check_parity(tra[:, :, 1023].cpu().numpy(), ot)
    for g in range(2):
        for k in range(4):
            check_parity(tcn[g][k][1023].cpu().numpy(), otcn[g][k], str((g, k)))


def _export(eng, st, N):
    conv = torch.zeros(2, N, 16, 6, 33, device="cuda")
    tra = torch.zeros(2, 3, N, 8, 2, device="cuda")
    tcn = [[torch.zeros(N, 16, 2 * d, 33, device="cuda") for d in (1, 2, 4, 8)] for _ in range(2)]
    eng.stream_export(st, conv, tra, tcn)
    return [conv, tra] + [t for grp in tcn for t in grp]


def test_single_launch_step_equals_the_three_launch_form(dev):
    """A single-frame step is ONE kernel (k_stream_ms: encoder -> both GTCN stacks -> decoder, nothing through HBM, the
    history of a block fetched one block ahead, only the new row written back).  With the stage taps enabled the
    library runs the three-launch form instead (its hand-off tensors are what the taps read): both forms must give
    the same output AND leave the same stream state, bit for bit, also when the last workgroup is partly filled."""
    from gtcrn_micro_amd import Engine
    p = load_params("rand")
    one, three = Engine(p, 0), Engine(p, 0)
    three.debug_enable(True)
    rng = np.random.default_rng(21)
    for N in (1, 7, 64):
        T = 19
        spec = cu((rng.standard_normal((N, 257, T, 2)) * 0.3).astype(np.float32))
        sa, sb_ = one.new_state(N), three.new_state(N)
        for t in range(T):
            ya = one.stream_step(sa, spec[:, :, t:t + 1])
            yb = three.stream_step(sb_, spec[:, :, t:t + 1])
            assert torch.equal(ya, yb), (N, t)
        assert torch.equal(sa, sb_), N                             # the whole ring state, counters included
        assert "k_stream_ms" not in three.timing_read()
    one.timing_enable(True)
    one.stream_step(sa, spec[:, :, :1])
    torch.cuda.synchronize()
    assert list(one.timing_read()) == ["k_stream_ms"]              # one launch per frame
    one.timing_enable(False)
    with pytest.raises(Exception):
        one.tap("en4", 0, 1)                                       # no hand-off tensors exist after a fused step


def test_stream_step_is_graph_capturable(dev):
    """gtcrn_stream_step issues no allocation and no synchronisation once the workspace is reserved: the per-frame call
    can be captured into a HIP graph and replayed frame after frame (new input copied into the captured buffer)."""
    from gtcrn_micro_amd import Engine
    eng = Engine(load_params("dns3"), 0)
    N, T = 16, 12
    gen = torch.Generator(device="cuda").manual_seed(7)
    spec = torch.randn(N, 257, T, 2, device="cuda", generator=gen) * 0.3
    eng.reserve(N, 1)
    ref_state = eng.new_state(N)
    ref = [eng.stream_step(ref_state, spec[:, :, t:t + 1]).clone() for t in range(T)]
    x = torch.empty(N, 257, 1, 2, device="cuda")
    y = torch.empty(N, 257, 1, 2, device="cuda")
    st = eng.new_state(N)
    warm = eng.new_state(N)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        x.copy_(spec[:, :, :1])
        eng.stream_step(warm, x, out=y)                            # warm-up on the capture stream (another state)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            eng.stream_step(st, x, out=y)
    st.copy_(eng.new_state(N))                                     # the capture itself did not run the step
    for t in range(T):
        x.copy_(spec[:, :, t:t + 1])
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, ref[t]), t
    assert torch.equal(st, ref_state)


def test_frame_counter_wraps_without_a_trace(dev):
    """The ring state carries a 16-bit frame counter (the reference's caches have none); only its value mod 16 (ring
    rows) and its parity matter.  A stream whose counter is poked to 65 530 and stepped across the wrap must behave
    exactly like the same stream at a counter congruent mod 16 -- outputs, exported caches and the counter after
    the wrap -- in the single-frame form and in a multi-frame call."""
    from gtcrn_micro_amd import Engine
    eng = Engine(load_params("rand"), 0)
    rng = np.random.default_rng(33)
    N, T0, T1 = 5, 10, 12                                          # 65530 mod 16 == 10
    spec = cu((rng.standard_normal((N, 257, T0 + T1, 2)) * 0.3).astype(np.float32))
    st = eng.new_state(N)
    for t in range(T0):
        eng.stream_step(st, spec[:, :, t:t + 1])
    assert st.view(torch.int32)[:, 0].tolist() == [T0] * N
    for chunks in ([1] * T1, [5, 7]):
        a, b_ = st.clone(), st.clone()
        b_.view(torch.int32)[:, 0] = 65530
        t0 = T0
        for c in chunks:
            ya = eng.stream_step(a, spec[:, :, t0:t0 + c])
            yb = eng.stream_step(b_, spec[:, :, t0:t0 + c])
            assert torch.equal(ya, yb), (chunks, t0)
            t0 += c
        assert a.view(torch.int32)[:, 0].tolist() == [T0 + T1] * N
        assert b_.view(torch.int32)[:, 0].tolist() == [(65530 + T1) & 0xFFFF] * N
        for ca, cb in zip(_export(eng, a, N), _export(eng, b_, N)):
            assert torch.equal(ca, cb), chunks
        b_.view(torch.int32)[:, 0] = a.view(torch.int32)[:, 0]
        assert torch.equal(a, b_)                                   # nothing else in the state differs


def _stream_module(tag="rand"):
    import json, os
    from conftest import GOLDEN
    from gtcrn_micro_amd.models.gtcrn_micro import GTCRNMicro
    from gtcrn_micro_amd.streaming.gtcrn_micro_stream import StreamGTCRNMicro
    from gtcrn_micro_amd.streaming.conversion.convert import convert_to_stream
    p = load_params(tag)
    man = json.load(open(os.path.join(GOLDEN, "params_manifest.json")))
    model = GTCRNMicro().eval()
    model.load_state_dict({n: torch.from_numpy(p[o:o + int(np.prod(s))].reshape(s).copy())
                           for n, s, o in man["tensors"]}, strict=False)
    sm = StreamGTCRNMicro().eval()
    convert_to_stream(sm, model)
    return sm.to("cuda")


def test_mirror_forward_fast_path_matches_reference_caches(dev):
    """The reference loop (gtcrn_micro_stream.py:626-635) passes the returned caches straight back in.  The mirror then
    skips the cache -> ring import (identity + version counters) and exports lazily: the caches read at frames 0, 1
    and 16 equal the reference's fixtures, with ONE import and three exports for 17 calls."""
    sm = _stream_module("rand")
    g = golden("stream_rand_T17.npz")
    spec = cu(g["spec"])
    conv_cache, tra_cache, tcn_cache = sm.init_caches(1, "cuda")
    ys = []
    with torch.no_grad():
        for i in range(17):
            y, conv_cache, tra_cache, tcn_cache = sm(spec[:, :, i:i + 1], conv_cache, tra_cache, tcn_cache)
            ys.append(y)
            if i in (0, 1, 16):                                    # reading a returned cache brings it up to date
                check_parity(conv_cache.cpu().numpy(), g[f"conv_cache_f{i}"], str(i))
                check_parity(tra_cache.cpu().numpy(), g[f"tra_cache_f{i}"], str(i))
                for gi in range(2):
                    for k in range(4):
                        check_parity(tcn_cache[gi][k].cpu().numpy(), g[f"tcn_cache_f{i}_g{gi}_b{k}"], str((i, gi, k)))
    check_parity(torch.cat(ys, 2).cpu().numpy(), g["spec_enh_stream"])
    assert sm.forward_stats == {"calls": 17, "imports": 1, "exports": 3}, sm.forward_stats


def test_mirror_forward_sees_caches_the_caller_modified(dev):
    """A caller that writes into a cache between two calls (here: resets ONE stream of three by zeroing its slices, as a
    server does when a call ends and a new one takes the slot) must get the reference's behaviour: that stream
    restarts from silence history, the others continue -- bit for bit what the native step gives for the same story.
    Also: a caller that keeps passing (and reading) its ORIGINAL tensors finds them current after every call."""
    from gtcrn_micro_amd import Engine
    sm = _stream_module("dns3")
    rng = np.random.default_rng(8)
    N, T, cut = 3, 20, 8
    spec = cu((rng.standard_normal((N, 257, T, 2)) * 0.3).astype(np.float32))
    # expected: native steps; stream 1's state is zeroed before frame `cut`
    eng = Engine(load_params("dns3"), 0)
    st = eng.new_state(N)
    want = []
    for t in range(T):
        if t == cut:
            st[1].zero_()
        want.append(eng.stream_step(st, spec[:, :, t:t + 1]))
    want = torch.cat(want, 2)
    conv_cache, tra_cache, tcn_cache = sm.init_caches(N, "cuda")
    got = []
    with torch.no_grad():
        for t in range(T):
            if t == cut:
                conv_cache[:, 1] = 0.0                             # in-place edits of the RETURNED caches
                tra_cache[:, :, 1] = 0.0
                for grp in tcn_cache:
                    for c in grp:
                        c[1] = 0.0
            y, conv_cache, tra_cache, tcn_cache = sm(spec[:, :, t:t + 1], conv_cache, tra_cache, tcn_cache)
            got.append(y)
    assert torch.equal(torch.cat(got, 2), want)
    assert sm.forward_stats["imports"] == 2 and sm.forward_stats["calls"] == T          # the first call and the edit
    # the caller's own tensors, passed and read every frame (the reference mutates them in place)
    sm2 = _stream_module("dns3")
    c0, t0, n0 = sm2.init_caches(N, "cuda")
    ref = eng.new_state(N)
    with torch.no_grad():
        for t in range(6):
            y, _, _, _ = sm2(spec[:, :, t:t + 1], c0, t0, n0)      # returned caches ignored on purpose
            eng.stream_step(ref, spec[:, :, t:t + 1])
            for a, b_ in zip([c0, t0] + [x for grp in n0 for x in grp], _export(eng, ref, N)):
                assert type(a) is torch.Tensor and torch.equal(a, b_), t
    assert sm2.forward_stats == {"calls": 6, "imports": 1, "exports": 6}


def test_mirror_forward_under_inference_mode_and_two_model_handover(dev):
    """infer.py runs under torch.inference_mode(): caches created there carry no version counter (reading
    ``t._version`` raises), so the mirror takes the plain import -> step -> export route for them -- same numbers as
    the native step, the caller's own tensors returned and current.  And a hand-over: model B receives the lazy caches
    model A handed out while A's export is still pending (an A/B of two checkpoints, a hot swap mid-stream): A writes
    them out first, so the stream continues exactly as a single native state fed by both models would."""
    from gtcrn_micro_amd import Engine
    rng = np.random.default_rng(21)
    N, T = 2, 9
    spec = cu((rng.standard_normal((N, 257, T, 2)) * 0.3).astype(np.float32))
    eng = Engine(load_params("dns3"), 0)
    st = eng.new_state(N)
    want = torch.cat([eng.stream_step(st, spec[:, :, t:t + 1]) for t in range(T)], 2)
    sm = _stream_module("dns3")
    with torch.inference_mode():
        conv_cache, tra_cache, tcn_cache = sm.init_caches(N, "cuda")
        assert conv_cache.is_inference()
        got = []
        for t in range(T):
            y, conv_cache, tra_cache, tcn_cache = sm(spec[:, :, t:t + 1], conv_cache, tra_cache, tcn_cache)
            got.append(y)
        assert type(conv_cache) is torch.Tensor
        assert torch.equal(torch.cat(got, 2), want)
        for a, b_ in zip([conv_cache, tra_cache] + [x for grp in tcn_cache for x in grp], _export(eng, st, N)):
            assert torch.equal(a, b_)
    assert sm.forward_stats == {"calls": T, "imports": T, "exports": T}
    # hand-over between two models with the same weights: frames 0..3 through A, 4..6 through B, 7..8 through A again
    smA, smB = _stream_module("dns3"), _stream_module("dns3")
    caches = list(smA.init_caches(N, "cuda"))
    got = []
    with torch.no_grad():
        for t in range(T):
            m = smB if 4 <= t < 7 else smA
            y, caches[0], caches[1], caches[2] = m(spec[:, :, t:t + 1], caches[0], caches[1], caches[2])
            got.append(y)
    assert torch.equal(torch.cat(got, 2), want)
    assert smA.forward_stats["imports"] == 2 and smB.forward_stats["imports"] == 1


def test_65536_streams_past_the_infinity_cache_equal_small_batches(dev):
    """bench.py's stream_capacity leg runs 65 536 (and more) concurrent streams per GPU: 10 GB of ring state, stream
    offsets past 2^31 bytes and 2^31 floats of state, 16 384 workgroups per launch.  Streams are independent, so any
    stream of the big batch must equal the same frames through a batch of four, bit for bit -- first, last and the
    ones either side of the 2^31-float boundary of the state tensor (stream 56 341)."""
    from gtcrn_micro_amd import Engine
    eng = Engine(load_params("dns3"), 0)
    N, T = 65536, 3
    gen = torch.Generator(device="cuda").manual_seed(65536)
    spec = (torch.randn(N, T, 257, 2, device="cuda", generator=gen) * 0.3).permute(0, 2, 1, 3)
    st = eng.new_state(N)
    assert st.numel() > 2 ** 31
    got = torch.cat([eng.stream_step(st, spec[:, :, t:t + 1]) for t in range(T)], 2)
    assert bool(torch.isfinite(got).all())
    for lo in (0, 56340, 65532):
        st4 = eng.new_state(4)
        want = torch.cat([eng.stream_step(st4, spec[lo:lo + 4, :, t:t + 1]) for t in range(T)], 2)
        assert torch.equal(got[lo:lo + 4], want), lo
        # ... and the ring state the big launch left for these streams is the small launch's
        assert torch.equal(st[lo:lo + 4], st4), lo
    del st, spec, got
    torch.cuda.empty_cache()


def test_stream_form_switch_gives_the_same_bits(dev):
    """gtcrn_stream_form: the one-launch step (default: form by stream count) against the three-launch form without the
    stage taps and against the one-launch step pinned to four (k_stream_ms) / seven (k_stream_wide) streams per workgroup
    -- the A/Bs of bench.py's stream_capacity leg: 70 streams (17 full workgroups of four + one of two; ten of seven), five
    frames -- outputs and the whole ring state bit-equal in every form; a wrong form is refused."""
    from gtcrn_micro_amd import Engine, GtcrnError
    eng = Engine(load_params("dns3"), 0)
    N, T = 70, 5
    gen = torch.Generator(device="cuda").manual_seed(70)
    spec = (torch.randn(N, T, 257, 2, device="cuda", generator=gen) * 0.3).permute(0, 2, 1, 3)
    st0 = eng.new_state(N)
    a = torch.cat([eng.stream_step(st0, spec[:, :, t:t + 1]) for t in range(T)], 2)
    for form in (1, 2, 3):
        st1 = eng.new_state(N)
        eng.stream_form(form)
        try:
            b = torch.cat([eng.stream_step(st1, spec[:, :, t:t + 1]) for t in range(T)], 2)
        finally:
            eng.stream_form(0)
        assert torch.equal(a, b) and torch.equal(st0, st1), form
    with pytest.raises(GtcrnError):
        eng.stream_form(4)


def test_staggered_many_round_launch_of_the_wide_step_changes_no_bit(dev):
    """From three rounds of workgroups on, the first round of a k_stream_wide launch starts phase-staggered (s_sleep before
    the first instruction: the CUs then do not all ask the HBM for their GTCN rows at once).  5 600 streams = 800 workgroups
    of seven: outputs and the whole ring state after six frames equal the four-streams-per-workgroup form's (no stagger there)
    bit for bit."""
    from gtcrn_micro_amd import Engine
    eng = Engine(load_params("dns3"), 0)
    N, T = 5600, 6
    gen = torch.Generator(device="cuda").manual_seed(56)
    spec = (torch.randn(N, T, 257, 2, device="cuda", generator=gen) * 0.3).permute(0, 2, 1, 3)
    res = []
    for form in (2, 3):
        st = eng.new_state(N)
        eng.stream_form(form)
        try:
            eng.timing_enable(True)
            out = torch.cat([eng.stream_step(st, spec[:, :, t:t + 1]) for t in range(T)], 2)
            torch.cuda.synchronize()
            names = list(eng.timing_read())
            eng.timing_enable(False)
        finally:
            eng.stream_form(0)
        res.append((out, st, names))
    assert res[0][2] == ["k_stream_ms"] and res[1][2] == ["k_stream_wide"]
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("which", ["dns3", "rand"])
def test_wide_single_launch_step_equals_the_narrow_one_and_offline(dev, which):
    """k_stream_wide (seven streams per workgroup, eight waves x two tiles, parameters streamed through LDS by DMA) against
    k_stream_ms (four per workgroup, parameters resident): 40 frames -- every TCN ring wraps, the dilation-8 ring more than
    twice -- at stream counts that leave the last workgroup with 1 .. 7 live streams; outputs and the whole ring state
    bit-equal, the wide form's name in the timing table, the streamed output == the offline forward of the same
    spectrogram bit for bit.  Then the form the library picks by itself: narrow at 1024 streams (one round of four-stream
    workgroups), wide at 1792 (two rounds of four-stream workgroups against one round of seven-stream ones)."""
    from gtcrn_micro_amd import Engine
    eng = Engine(load_params(which), 0)
    T = 40
    for N in (1, 6, 7, 8, 13, 30, 100):
        gen = torch.Generator(device="cuda").manual_seed(1000 + N)
        spec = (torch.randn(N, T, 257, 2, device="cuda", generator=gen) * 0.3).permute(0, 2, 1, 3)
        outs, states = [], []
        for form in (2, 3):
            st = eng.new_state(N)
            eng.stream_form(form)
            try:
                outs.append(torch.cat([eng.stream_step(st, spec[:, :, t:t + 1]) for t in range(T)], 2))
            finally:
                eng.stream_form(0)
            states.append(st)
        assert torch.equal(outs[0], outs[1]), N
        assert torch.equal(states[0], states[1]), N
        assert torch.equal(outs[1], eng.forward_spec(spec.contiguous())), N
    for N, name in ((1024, "k_stream_ms"), (1792, "k_stream_wide")):
        eng.timing_enable(True)
        st = eng.new_state(N)
        spec = torch.zeros(N, 257, 1, 2, device="cuda")
        eng.stream_step(st, spec)
        torch.cuda.synchronize()
        assert list(eng.timing_read()) == [name], (N, eng.timing_read())
    eng.timing_enable(False)
