"""CPU-side checks of the product's host logic: the C ABI loads and exports every declared
symbol, the parameter table matches the reference state_dict, and the packed slot-space
buffers reproduce the oracle when run through a numpy emulation of the kernels' dataflow."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden, load_params, rel_err

import __graft_entry__ as graft


@pytest.fixture(scope="module", autouse=True)
def _built():
    graft.build()


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gtcrn_micro_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(gtcrn_\w+)\s*\(", hdr))
    assert len(names) >= 28
    L = ctypes.CDLL(os.path.join(ROOT, "gtcrn_micro_amd", "libgtcrn_micro_hip.so"))
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_one_launch_streaming_form_is_picked_by_rounds_of_workgroups():
    """gtcrn_stream_streams_per_workgroup (host logic, no device): both one-launch forms of a single-frame step run one
    workgroup per CU, so a step costs rounds-of-256-workgroups x the form's time per round; four streams per workgroup
    (k_stream_ms) while that needs no more rounds than seven (k_stream_wide, 1.5 x the time per round) would, seven from
    there on.  BASELINE configs[2]'s own 1 024 streams stay one round of the narrow form."""
    from gtcrn_micro_amd import _lib, GtcrnError
    f = _lib.stream_streams_per_workgroup
    assert [f(n) for n in (1, 4, 1000, 1024)] == [4, 4, 4, 4]               # one round either way: the faster round
    assert f(1025) == 7 and f(1792) == 7                                    # two rounds of four against ONE of seven
    assert f(2048) == 4                                                     # 512 workgroups of four = 2 rounds; 293 of seven = 2
    assert f(4096) == 4                                                     # 4 rounds against 3 x 1.5
    assert all(f(n) == 7 for n in (5376, 8192, 16384, 65536, 262144))
    for n in range(1, 20000, 37):                                           # the rule itself
        r4, r7 = -(-(-(-n // 4)) // 256), -(-(-(-n // 7)) // 256)
        assert f(n) == (7 if r7 * 1.5 < r4 else 4), n
    with pytest.raises(GtcrnError):
        f(0)


def test_param_table_matches_reference_state_dict():
    from gtcrn_micro_amd import _lib
    man = json.load(open(os.path.join(GOLDEN, "params_manifest.json")))
    table = _lib.param_table()
    assert len(table) == len(man["tensors"]) == 342
    for (name, numel, off), (rname, rshape, roff) in zip(table, man["tensors"]):
        assert name == rname and numel == int(np.prod(rshape)) and off == roff
    assert table[-1][1] + table[-1][2] == man["n_floats"] == _lib.NPARAM_FLOATS


def test_module_mirror_state_dict_and_errors():
    import torch
    from gtcrn_micro_amd.models.gtcrn_micro import GTCRNMicro, erb_filter_bank, state_dict_to_blob
    from gtcrn_micro_amd import GtcrnError
    p = load_params("dns3")
    assert np.array_equal(erb_filter_bank().ravel(), p[:64 * 192])      # fixed ERB bank, bit for bit
    m = GTCRNMicro(n_fft=512, hop_len=256, win_len=512)
    man = json.load(open(os.path.join(GOLDEN, "params_manifest.json")))
    sd = m.state_dict()
    assert len(sd) == 388
    # load the shipped checkpoint blob by name and get the same blob back
    new = {}
    for name, shape, off in man["tensors"]:
        new[name] = torch.from_numpy(p[off:off + int(np.prod(shape))].reshape(shape).copy())
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            new[k] = v
    m.load_state_dict({"module." + k: v for k, v in new.items()})        # DDP prefix tolerated
    assert np.array_equal(state_dict_to_blob(m.state_dict()), p)
    m.eval()
    with pytest.raises(GtcrnError):
        m(torch.zeros(1, 257, 4, 2))                                      # CPU tensor: no CPU path
    m.train()
    with pytest.raises(GtcrnError):
        m(torch.zeros(1, 257, 4, 2))                                      # train mode has no CPU path either


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gtcrn_micro_amd import Engine, GtcrnError, Trainer
    with pytest.raises(GtcrnError):
        Engine(load_params("dns3"), 0)
    with pytest.raises(GtcrnError):
        Trainer(0)                                   # the training path has no CPU fallback either


def test_loss_mirror_has_no_cpu_path():
    import torch
    from gtcrn_micro_amd import GtcrnError
    from gtcrn_micro_amd.loss import HybridLoss
    with pytest.raises(GtcrnError):
        HybridLoss()(torch.zeros(1, 257, 4, 2), torch.zeros(1, 257, 4, 2))


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pat = re.compile(r"^\s*(from|import)\s+oracle|#include\s+\".*oracle|libgtcrn_oracle|oracle\.py", re.M)
    for top in ("gtcrn_micro_amd", "tools", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, top)):
            for f in fs:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".sh")):
                    txt = open(os.path.join(dp, f)).read()
                    assert not pat.search(txt), os.path.join(dp, f)


@pytest.mark.parametrize("tag", ["dns3", "rand"])
def test_packed_slot_space_reproduces_reference(tag):
    from gtcrn_micro_amd import _lib
    import slot_emulator as E
    g = golden(f"offline_{tag}_T17.npz")
    F, I = _lib.pack_params_host(load_params(tag))
    rec = E.forward(F, I, g["spec"][0])
    for name in ("en0", "en1", "en2", "en3", "en4", "de0", "de1", "de2", "de3"):
        assert rel_err(E.to_logical(name, rec[name], I), g[name][0]) < 2e-5, name
    assert rel_err(E.to_logical("gtcn1", rec["gtcn1"], I), g["gtcn1_b3"][0]) < 2e-5
    assert rel_err(E.to_logical("gtcn2", rec["gtcn2"], I), g["gtcn2_b3"][0]) < 2e-5
    assert rel_err(rec["de4"], g["de4"][0]) < 2e-5
    assert rel_err(rec["out"], g["spec_enh"][0]) < 2e-5


def test_make_window_close_to_torch():
    from gtcrn_micro_amd import make_window
    g = golden("offline_dns3_T17.npz")
    assert np.abs(make_window(0) - g["window"]).max() < 2e-6


def test_checkpoint_dict_round_trip(tmp_path):
    """The reference's checkpoint layout (train.py:200-216): epoch/optimizer/scheduler/model, 388 model keys;
    CPU-only (parameter containers), no kernels involved."""
    import torch
    from gtcrn_micro_amd.models.gtcrn_micro import GTCRNMicro, state_dict_to_blob
    from gtcrn_micro_amd.train import load_checkpoint, save_checkpoint
    from gtcrn_micro_amd.utils.scheduler import LinearWarmupCosineAnnealingLR
    torch.manual_seed(3)
    m = GTCRNMicro()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = LinearWarmupCosineAnnealingLR(opt, 25000, 250000, 1e-3, 1e-6)
    assert abs(opt.param_groups[0]["lr"]) < 1e-12               # step 0 of the warm-up
    for p in m.parameters():
        if p.requires_grad:
            p.grad = torch.ones_like(p) * 0.01
    opt.step(); sch.step()
    assert abs(opt.param_groups[0]["lr"] - 4e-8) < 1e-15        # lr after step 1 = 4e-8
    path = str(tmp_path / "model_007.tar")
    save_checkpoint(path, m, opt, sch, 7)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert sorted(ck) == ["epoch", "model", "optimizer", "scheduler"] and len(ck["model"]) == 388
    m2 = GTCRNMicro()
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    sch2 = LinearWarmupCosineAnnealingLR(opt2, 25000, 250000, 1e-3, 1e-6)
    assert load_checkpoint(path, m2, opt2, sch2) == 8
    assert np.array_equal(state_dict_to_blob(m2.state_dict()), state_dict_to_blob(m.state_dict()))
    st, st2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert sch2.last_epoch == 1 and sorted(st) == sorted(st2) and len(st) == 248
    k0 = sorted(st)[0]
    assert float(st[k0]["step"]) == float(st2[k0]["step"]) == 1.0
    # DDP-prefixed checkpoints load too
    ck["model"] = {"module." + k: v for k, v in ck["model"].items()}
    torch.save(ck, path)
    assert load_checkpoint(path, GTCRNMicro()) == 8


def test_packer_under_address_and_ub_sanitizers(tmp_path):
    """csrc/pack.cpp (the only non-trivial host arithmetic of the product) runs clean under ASan + UBSan."""
    import shutil
    import subprocess
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "asan_pack_driver")
    csrc = os.path.join(ROOT, "gtcrn_micro_amd", "csrc")
    r = subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I", csrc, "-o", exe, os.path.join(ROOT, "tests", "asan_pack_driver.cpp"),
                        os.path.join(csrc, "pack.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for tag in ("dns3", "rand"):
        r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", f"params_{tag}.f32")], capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0 and "asan_pack_driver ok" in r.stdout, (r.returncode, r.stderr[-3000:])


def test_convert_to_stream_weight_contract_on_cpu():
    """convert_to_stream is a state-dict remap (no arithmetic): every wrapper infix resolves, a ConvTranspose2d
    weight becomes W'[o,i,a,b] = W[i,o,kT-1-a,kF-1-b] (convert.py:35-48, frozen in conv_wrappers.npz from the
    reference's own function), an unmatched key raises ValueError("Key error!") (:54)."""
    import torch
    import torch.nn as nn
    from conftest import golden
    from gtcrn_micro_amd.streaming.conversion.convert import convert_to_stream
    from gtcrn_micro_amd.streaming.conversion.convolution import StreamConv1d, StreamConv2d, StreamConvTranspose2d
    g = golden("conv_wrappers.npz")
    de = nn.ConvTranspose2d(4, 8, (3, 1), stride=(1, 1), padding=(4, 1), dilation=(2, 2))
    with torch.no_grad():
        de.weight.copy_(torch.from_numpy(g["ct2d_w"])); de.bias.copy_(torch.from_numpy(g["ct2d_b"]))
    sde = StreamConvTranspose2d(4, 8, (3, 1), stride=(1, 1), padding=(0, 1), dilation=(2, 2))
    convert_to_stream(sde, de)
    assert np.array_equal(sde.ConvTranspose2d.weight.detach().numpy(), g["ct2d_w_stream"])
    assert np.array_equal(sde.ConvTranspose2d.bias.detach().numpy(), g["ct2d_b"])
    c2, s2 = nn.Conv2d(2, 3, 3), StreamConv2d(2, 3, 3)
    convert_to_stream(s2, c2)
    assert torch.equal(s2.Conv2d.weight, c2.weight) and torch.equal(s2.Conv2d.bias, c2.bias)
    c1, s1 = nn.Conv1d(4, 4, 3, groups=4), StreamConv1d(4, 4, 3, groups=4)
    convert_to_stream(s1, c1)
    assert torch.equal(s1.Conv1d.weight, c1.weight)
    with pytest.raises(ValueError, match="Key error!"):
        convert_to_stream(s2, nn.Sequential(nn.Conv2d(2, 3, 3)))      # keys '0.weight', '0.bias'


REF_CKPT = os.path.join(GOLDEN, "ckpt_ref", "model_002.tar")


def test_reference_written_checkpoint_loads(tmp_path):
    """A checkpoint written by the REFERENCE's own Trainer._save_checkpoint (train.py:200-221; fixture generator
    tests/golden/make_golden_ckpt.py) resumes here: epoch, Adam state, schedule position and all 388 model keys
    (counterpart of Trainer._resume_checkpoint, :223-237); written back with save_checkpoint it is the same dict."""
    import torch
    from gtcrn_micro_amd.models.gtcrn_micro import GTCRNMicro
    from gtcrn_micro_amd.train import load_checkpoint, save_checkpoint
    from gtcrn_micro_amd.utils.scheduler import LinearWarmupCosineAnnealingLR
    io = np.load(os.path.join(GOLDEN, "ckpt_ref_io.npz"))
    ck = torch.load(REF_CKPT, map_location="cpu", weights_only=False)
    assert sorted(ck) == ["epoch", "model", "optimizer", "scheduler"] and len(ck["model"]) == 388 == int(io["n_keys"])
    m = GTCRNMicro()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = LinearWarmupCosineAnnealingLR(opt, 25000, 250000, 1e-3, 1e-6)
    assert load_checkpoint(REF_CKPT, m, opt, sch) == 3                 # start_epoch = epoch + 1
    sd = m.state_dict()
    assert list(sd) == list(ck["model"])                              # same keys, same order
    assert all(torch.equal(sd[k], ck["model"][k]) for k in sd)
    assert abs(opt.param_groups[0]["lr"] - float(io["lr"])) < 1e-15 and sch.last_epoch == 2
    st = opt.state_dict()["state"]
    assert len(st) == 248 and all(float(v["step"]) == float(io["adam_step"]) for v in st.values())
    ref_st = ck["optimizer"]["state"]
    assert all(torch.equal(st[i]["exp_avg"], ref_st[i]["exp_avg"]) and
               torch.equal(st[i]["exp_avg_sq"], ref_st[i]["exp_avg_sq"]) for i in ref_st)
    # write it back: the reference's reader (torch.load + the same four keys) sees identical content
    out = str(tmp_path / "model_003.tar")
    save_checkpoint(out, m, opt, sch, 2)
    ck2 = torch.load(out, map_location="cpu", weights_only=False)
    assert ck2["epoch"] == ck["epoch"] and list(ck2["model"]) == list(ck["model"])
    assert all(torch.equal(ck2["model"][k], ck["model"][k]) for k in ck["model"])
    assert ck2["scheduler"]["last_epoch"] == ck["scheduler"]["last_epoch"]
    assert ck2["optimizer"]["param_groups"][0]["lr"] == ck["optimizer"]["param_groups"][0]["lr"]


def test_infer_host_helpers(tmp_path):
    """Host side of the bulk driver: 8-bit PCM is unsigned with offset 128; the per-rank scp lists merge in rank order."""
    from scipy.io import wavfile
    from gtcrn_micro_amd.infer import extract_fileid, merge_scp, read_wav_f32
    wavfile.write(tmp_path / "u8.wav", 16000, np.array([0, 128, 255, 64], np.uint8))
    fs, x = read_wav_f32(str(tmp_path / "u8.wav"))
    assert fs == 16000 and np.allclose(x, [-1.0, 0.0, 127 / 128, -0.5])
    assert extract_fileid("/a/b/x_snr5_fileid_12.wav") == "12" and extract_fileid("nope.wav") is None
    for r, rows in enumerate((["a 1", "b 2"], ["c 3"])):
        for f in ("inf.scp", "ref.scp"):
            (tmp_path / f"{f}.rank{r}").write_text("".join(x + "\n" for x in rows))
    merge_scp(str(tmp_path), 2)
    assert (tmp_path / "inf.scp").read_text() == "a 1\nb 2\nc 3\n" == (tmp_path / "ref.scp").read_text()


def test_quantised_weight_contract_on_host():
    """configs[4] weight contract without a GPU: the packer's quantised buffers hold fp16(int8 * per-output-channel
    scale) -- checked against the independent restatement in oracle/quant_port.py for the dense 3x3, a pointwise,
    a depthwise and the ERB tables -- and the host float -> half rounding equals numpy's."""
    import torch
    from gtcrn_micro_amd import _lib
    from oracle.quant_port import QuantPort
    p = load_params("rand")
    F, I = _lib.pack_params_host(p)
    Fq, Iq = _lib.pack_params_host(p, quant=True)
    assert np.array_equal(I, Iq) and not np.array_equal(F, Fq)
    port = QuantPort(p)
    # decoder block 0, dense transposed 3x3: packed tap (kt,kf) row o col i == W'[i,o,kt,kf] (input slots of de0 are
    # the gtcn slot order: compare as multisets per output row, which is what the per-row scale acts on)
    wq = port.w["decoder.de_convs.0.depth_conv"].numpy()                      # (in,out,3,3)
    # float offsets of csrc/layout.h
    P_ENC, ENC_SIZE, GTCN_SIZE = 0, 768 + 12 + 276 + 1300 + 3 * 828, 4 * 612
    P_DEC = P_ENC + ENC_SIZE + 2 * GTCN_SIZE
    GB_DN_A = 828
    for o in range(16):
        packed = np.sort(np.concatenate([Fq[P_DEC + GB_DN_A + t * 256 + o * 16: P_DEC + GB_DN_A + t * 256 + o * 16 + 16]
                                         for t in range(9)]))
        assert np.array_equal(packed, np.sort(wq[:, o].ravel())), o
        assert np.array_equal(packed, packed.astype(np.float16).astype(np.float32))        # fp16 values
        assert not np.array_equal(packed, np.sort(np.concatenate(
            [F[P_DEC + GB_DN_A + t * 256 + o * 16: P_DEC + GB_DN_A + t * 256 + o * 16 + 16] for t in range(9)])))
    # the same matrices as fp16 K-chunks for v_mfma_f32_16x16x32_f16 (chunk c = taps 2c | 2c+1; written over the head
    # of the block's D_DN16 slot in the quantised buffer): exactly the quantised fp32 values, zero in the spare half
    from tests.slot_emulator import K as KL, _mat as unsw
    for j in range(3):
        for c in range(5):
            m = unsw(Fq, P_DEC + KL["D_DN16"] + j * KL["DN16_SIZE"] + c * 256)
            hv = np.ascontiguousarray(m).view(np.float16).reshape(16, 32).astype(np.float32)
            for half in range(2):
                tap = 2 * c + half
                want = (unsw(Fq, P_DEC + KL["D_BLK"] + j * KL["GBD_SIZE"] + KL["GB_DN_A"] + tap * 256) if tap < 9
                        else np.zeros((16, 16), np.float32))
                assert np.array_equal(hv[:, half * 16:half * 16 + 16], want), (j, c, half)
    # en_conv1 (Conv2d 16->16, 5 taps, identity slot order): E_EN1_A[k][o][i] == W'[o,i,0,k]
    E_EN1_A = 768 + 12 + 276
    w1 = port.w["encoder.en_convs.1.conv"].numpy()
    for k in range(5):
        from tests.slot_emulator import _mat
        assert np.array_equal(_mat(Fq, E_EN1_A + k * 256), w1[:, :, 0, k])      # (rows stored half-swapped: _mat undoes it)
    # ERB.bm bands: the nonzeros of row j
    erb = port.w["erb"].numpy()
    for j in (0, 17, 63):
        lo, cnt = int(I[j]), int(I[64 + j])                   # I_ERB_LO, I_ERB_N: the band's support in the fp32 bank
        assert cnt > 0 and np.array_equal(Fq[j * 12: j * 12 + cnt], erb[j, lo:lo + cnt])
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(2000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 1e3, 7e4)] +
                        [np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e6, 6.1e-5, 5.96e-8, 2.98e-8, 2.9e-8], np.float32)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).astype(np.float32)
    got = np.array([_lib.round_to_half(float(x)) for x in xs], np.float32)
    assert np.array_equal(got, want)


def test_quality_scores_closed_form():
    """quant.sisnr_metric / sdr_metric restate eval_intrusive_metrics.py:74-91: closed-form cases."""
    from gtcrn_micro_amd.quant import sdr_metric, sisnr_metric
    rng = np.random.default_rng(1)
    ref = rng.standard_normal(16000)
    noise = rng.standard_normal(16000)
    noise -= noise.mean()
    r0 = ref - ref.mean()
    noise -= (noise @ r0) / (r0 @ r0) * r0                   # exactly orthogonal to the (mean-removed) reference
    for snr in (0.0, 10.0, 30.0):
        g = np.sqrt((r0 @ r0) / (noise @ noise) / 10 ** (snr / 10))
        inf = ref + g * noise
        assert abs(sisnr_metric(ref, inf) - snr) < 1e-6 and abs(sdr_metric(ref, inf) - snr) < 1e-6
        assert abs(sisnr_metric(ref, 0.5 * inf + 3.0) - snr) < 1e-6          # scale and offset invariant
        want = 10 * np.log10(1.0 / (0.25 + 0.25 / 10 ** (snr / 10)))       # SDR is not: residual = -ref/2 + noise/2
        assert abs(sdr_metric(ref, 0.5 * inf) - want) < 1e-6
    assert sdr_metric(ref, ref) > 100


def test_headline_kernels_use_no_scratch():
    """The headline kernels sit at the 168-VGPR limit of three waves per SIMD; a spill costs s_waitcnt vmcnt(0) on every
    reload and a harmless-looking edit can tip them over (the decoder went 0.52 -> 0.60 ms that
    way in round 2).  Compile-only check (hipcc cross-compiles without a GPU)."""
    import subprocess
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh")], capture_output=True, text=True,
                         timeout=900).stdout
    rows = {l.split("|")[0].strip(): l for l in out.splitlines() if "ScratchSize" in l}
    must = ["void gtk::k_front<true, false>", "void gtk::k_encoder<3, false, false, false, false>",
            "void gtk::k_gtcn_band<false, false>", "void gtk::k_decoder<false, 3, false, false, false>", "gtk::k_istft",
            "gtk::k_gtcn_ms", "void gtk::k_encoder<1, true, false, true, false>",
            "void gtk::k_decoder<false, 1, true, false, false>",
            # chunked-streaming forms (in-kernel front end), one / two / three tiles per wave
            "void gtk::k_encoder<3, false, false, true, false>", "void gtk::k_encoder<1, false, false, true, false>",
            "void gtk::k_decoder<false, 1, false, false, false>", "void gtk::k_decoder<false, 2, false, false, false>",
            # time-span forms (offline batches that do not fill the 256 CUs)
            "void gtk::k_encoder<3, false, false, false, true>", "void gtk::k_gtcn_band<false, true>"]
    must.append("gtk::k_stream_ms")
    for k in must:
        assert k in rows, (k, sorted(rows))
        assert "ScratchSize [bytes/lane]: 0 " in rows[k], rows[k]
    # ... and so does every other kernel of the product path.  Exceptions: the stage-tap instantiations of the decoder
    # (k_decoder<true, ...>, used by the parity tests alone), and the time-span form of the decoder, which parks ONE
    # 16-byte value in scratch before its segment loop and reloads it where a share runs on into the next utterance --
    # outside the chunk loop (at most once per workgroup).  Its budget is pinned so that it cannot grow unnoticed.
    span_dec = "void gtk::k_decoder<false, 3, false, false, true>"
    assert span_dec in rows
    import re
    assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", rows[span_dec]).group(1)) <= 32, rows[span_dec]
    for k, row in rows.items():
        if not k.startswith("void gtk::k_decoder<true") and k != span_dec:
            assert "ScratchSize [bytes/lane]: 0 " in row, row


def test_dense_bf16_planes_recombine_to_the_fp32_matrices():
    """The decoder's dense 3x3 runs on v_mfma_f32_16x16x32_bf16 from three bf16 planes per weight (layout.h D_DN16):
    the planes must add up EXACTLY to the folded fp32 slot matrices (hi + mid + lo == w, every residual exact), sit
    in the K order the kernel reads (chunk c = taps 2c, 2c+1; k = tap half * 16 + hidden channel), and the unused
    half of the last chunk must be zero."""
    from gtcrn_micro_amd import _lib
    from tests.slot_emulator import K, _mat
    for tag in ("dns3", "rand"):
        F, _ = _lib.pack_params_host(load_params(tag))
        P_DEC = K["P_DEC"]
        for j in range(3):
            for c in range(K["DN16_CHUNKS"]):
                planes = []
                for p in range(3):
                    off = P_DEC + K["D_DN16"] + j * K["DN16_SIZE"] + (c * 3 + p) * 256
                    m = _mat(F, off)                                   # undo the row swizzle on the 16x16-float view
                    bits = np.ascontiguousarray(m).view(np.uint16).reshape(16, 32).astype(np.uint32) << 16
                    planes.append(bits.view(np.float32).astype(np.float64))
                tot = planes[0] + planes[1] + planes[2]                # exact in float64
                assert np.all(np.abs(planes[1]) <= np.abs(planes[0]) * 2.0 ** -8 + 1e-45)
                assert np.all(np.abs(planes[2]) <= np.abs(planes[0]) * 2.0 ** -16 + 1e-45)
                for half in range(2):
                    tap = 2 * c + half
                    got = tot[:, half * 16:half * 16 + 16]
                    if tap < 9:
                        want = _mat(F, P_DEC + K["D_BLK"] + j * K["GBD_SIZE"] + K["GB_DN_A"] + tap * 256)
                        assert np.array_equal(got, want.astype(np.float64)), (tag, j, c, half)
                    else:
                        assert not got.any()


def test_replaced_parameter_object_is_noticed():
    """the cached list of tensors behind the state_dict must not go stale when a Parameter or buffer OBJECT
    is replaced (`mod.weight = nn.Parameter(...)`, parametrize, weight_norm, `register_buffer`): no version counter
    moves then, so the weight signature carries a serial that the registration hooks advance."""
    import torch
    from gtcrn_micro_amd.models.gtcrn_micro import GTCRNMicro
    m = GTCRNMicro()
    sig0 = m._signature()
    assert m._signature() == sig0                                # stable while nothing changes
    other = torch.nn.Linear(3, 3)                                # registrations elsewhere do not disturb it
    assert m._signature() == sig0 and other is not None
    conv = m.encoder.en_convs[0].conv
    old = conv.weight
    conv.weight = torch.nn.Parameter(torch.full_like(old, 0.25))  # a NEW object, version 0 like the old one
    sig1 = m._signature()
    assert sig1 != sig0
    assert any(t is conv.weight for t in m._state_tensors()[0]) and not any(t is old for t in m._state_tensors()[0])
    bn = m.encoder.en_convs[0].bn
    bn.register_buffer("running_mean", torch.ones(16))
    sig2 = m._signature()
    assert sig2 != sig1
    with torch.no_grad():
        conv.weight.mul_(2.0)                                    # in-place edits still move the version sum
    assert m._signature()[0] == sig2[0] + 1 and m._signature()[2] == sig2[2]


def test_folder_driver_exchanges_a_failure_flag_before_merging(tmp_path, monkeypatch):
    """a per-file error on one rank must not leave the other ranks in a collective, and a half-finished run
    must not be merged.  The per-rank work runs first, then `agree(ok)` is called on EVERY rank -- also on the one that
    failed -- and only then the failing rank re-raises, the others refuse to merge."""
    from gtcrn_micro_amd import infer
    calls = []

    def agree_all(ok):
        calls.append(ok)
        return ok and agree_all.others_ok
    monkeypatch.setattr(infer, "merge_scp", lambda d, w: calls.append("merge"))
    # (1) this rank fails: the exchange still happens, then its own error surfaces
    monkeypatch.setattr(infer, "_enhance_shard", lambda *a, **k: (_ for _ in ()).throw(AssertionError("x.wav: sample rate 8000 != 16000")))
    agree_all.others_ok = True
    with pytest.raises(AssertionError, match="sample rate"):
        infer.enhance_folder("n", "c", str(tmp_path), "ck", rank=1, world=2, agree=agree_all)
    assert calls == [False]
    # (2) this rank is fine, another one failed: no merge, non-zero exit
    calls.clear()
    monkeypatch.setattr(infer, "_enhance_shard", lambda *a, **k: ([("u", "p")], [("u", "r")]))
    agree_all.others_ok = False
    with pytest.raises(RuntimeError, match="another rank failed"):
        infer.enhance_folder("n", "c", str(tmp_path), "ck", rank=0, world=2, agree=agree_all)
    assert calls == [True]
    # (3) everybody is fine: rank 0 merges, the others do not
    for rank, want in ((0, [True, "merge"]), (1, [True])):
        calls.clear()
        agree_all.others_ok = True
        assert infer.enhance_folder("n", "c", str(tmp_path), "ck", rank=rank, world=2, agree=agree_all)[0] == [("u", "p")]
        assert calls == want
    # --device with a multi-process launch puts every rank on one GPU: refused
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    with pytest.raises(SystemExit):
        infer.main(["--noisy-dir", "n", "--clean-dir", "c", "--enh-dir", str(tmp_path), "--checkpoint", "ck", "-D", "0"])


def test_stream_mirror_cache_binding_logic_without_a_gpu(monkeypatch):
    """The reference-shaped streaming call hands caller-owned caches back and forth (gtcrn_micro_stream.py:626-635).
    The mirror skips the cache -> ring import while the caller passes back what it was handed and has not written to
    it, and defers the ring -> cache export until a returned cache is actually read.  Host logic only: the engine is
    replaced by a recorder whose "state" is a frame counter that import reads from / export writes to the caches."""
    import torch
    from gtcrn_micro_amd.streaming import gtcrn_micro_stream as M

    class FakeEngine:
        def __init__(self):
            self.log = []

        def new_state(self, n):
            return torch.zeros(n, 1)

        def stream_import(self, state, conv, tra, tcn):
            self.log.append("import")
            state[:, 0] = conv[0, :, 0, 0, 0]                    # the "state" of a stream lives in one cache element

        def stream_export(self, state, conv, tra, tcn):
            self.log.append("export")
            conv[0, :, 0, 0, 0] = state[:, 0]

        def stream_step(self, state, spec):
            self.log.append("step")
            state += 1.0
            return spec * 2.0

    sm = M.StreamGTCRNMicro().eval()
    eng = FakeEngine()
    monkeypatch.setattr(M.StreamGTCRNMicro, "engine", lambda self, dev: eng)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))          # no CPU path in the real one
    B = 3
    caches = list(sm.init_caches(B, "cpu"))
    orig = [caches[0], caches[1]] + [t for grp in caches[2] for t in grp]
    x = torch.ones(B, 257, 1, 2)
    # (1) the reference loop: returned caches straight back in -> one import, one eager export (fresh binding), no more
    for t in range(5):
        y, caches[0], caches[1], caches[2] = sm(x, caches[0], caches[1], caches[2])
    assert eng.log == ["import", "step", "export"] + ["step"] * 4
    assert isinstance(caches[0], M._LazyCache) and caches[0].data_ptr() == orig[0].data_ptr()
    # reading a returned cache brings it (and the caller's own storage) up to date, once
    assert float(caches[0][0, 1, 0, 0, 0]) == 5.0 and eng.log[-1] == "export" and float(orig[0][0, 1, 0, 0, 0]) == 5.0
    n = len(eng.log)
    assert float(caches[0][0, 2, 0, 0, 0]) == 5.0 and len(eng.log) == n                  # nothing pending: no second export
    assert caches[0].shape == (2, B, 16, 6, 33) and len(eng.log) == n                   # metadata never triggers it
    # (2) an in-place edit between two calls is seen: export first (the edit lands on current data), then re-import
    y, caches[0], caches[1], caches[2] = sm(x, caches[0], caches[1], caches[2])
    caches[0][0, 1, 0, 0, 0] = 100.0
    assert eng.log[-2:] == ["step", "export"]
    y, caches[0], caches[1], caches[2] = sm(x, caches[0], caches[1], caches[2])
    assert eng.log[-3:] == ["import", "step", "export"]
    assert [float(v) for v in caches[0][0, :, 0, 0, 0]] == [7.0, 101.0, 7.0]
    # (3) a caller that keeps its ORIGINAL tensors: no import while untouched, but an export every frame
    sm2 = M.StreamGTCRNMicro().eval()
    eng.log.clear()
    c0, t0, n0 = sm2.init_caches(B, "cpu")
    for t in range(3):
        sm2(x, c0, t0, n0)
        assert type(c0) is torch.Tensor and float(c0[0, 0, 0, 0, 0]) == t + 1.0
    assert eng.log == ["import", "step", "export", "step", "export", "step", "export"]
    c0[0, 0, 0, 0, 0] = -5.0                                                          # the caller writes its own tensor
    sm2(x, c0, t0, n0)
    assert eng.log[-3:] == ["import", "step", "export"] and float(c0[0, 0, 0, 0, 0]) == -4.0
    # (4) wrong cache shapes raise the reference's AssertionError before anything runs
    with pytest.raises(AssertionError):
        sm2(x, c0[:, :, :, :4], t0, n0)
    # (5) a raw pointer is a read: data_ptr() of a returned cache brings the memory up to date first (a caller that
    #     hands the pointer to its own kernels must not see stale contents); so do __dlpack__ and deepcopy
    import copy
    sm3 = M.StreamGTCRNMicro().eval()
    eng.log.clear()
    caches = list(sm3.init_caches(B, "cpu"))
    own = caches[0]
    for t in range(3):
        y, caches[0], caches[1], caches[2] = sm3(x, caches[0], caches[1], caches[2])
    assert eng.log[-1] == "step" and float(own.as_subclass(torch.Tensor)[0, 0, 0, 0, 0]) == 1.0    # stale: export pending
    ptr = caches[0].data_ptr()
    assert eng.log[-1] == "export" and ptr == own.data_ptr() and float(own[0, 0, 0, 0, 0]) == 3.0
    y, caches[0], caches[1], caches[2] = sm3(x, caches[0], caches[1], caches[2])
    dup = copy.deepcopy(caches[0])
    assert type(dup) is torch.Tensor and float(dup[0, 0, 0, 0, 0]) == 4.0 and dup.data_ptr() != ptr
    y, caches[0], caches[1], caches[2] = sm3(x, caches[0], caches[1], caches[2])
    assert eng.log[-1] == "step"
    torch.from_dlpack(caches[1])                                                      # calls caches[1].__dlpack__()
    assert eng.log[-1] == "export"
    # (6) caches created under torch.inference_mode() (infer.py runs in it) track no version counter: the plain route,
    #     import -> step -> export on every call, the caller's own tensors returned; edits are always seen
    sm4 = M.StreamGTCRNMicro().eval()
    eng.log.clear()
    with torch.inference_mode():
        ci, ti, ni = sm4.init_caches(B, "cpu")
        assert ci.is_inference()
        for t in range(3):
            y, ci2, ti2, ni2 = sm4(x, ci, ti, ni)
            assert ci2 is ci and type(ci2) is torch.Tensor and float(ci[0, 0, 0, 0, 0]) == t + 1.0
        ci[0, 0, 0, 0, 0] = 40.0
        sm4(x, ci, ti, ni)
        assert float(ci[0, 0, 0, 0, 0]) == 41.0
    assert eng.log == ["import", "step", "export"] * 4
    # ... while ordinary caches handed through inference_mode keep the fast path, edits included
    sm5 = M.StreamGTCRNMicro().eval()
    eng.log.clear()
    caches = list(sm5.init_caches(B, "cpu"))
    with torch.inference_mode():
        for t in range(4):
            y, caches[0], caches[1], caches[2] = sm5(x, caches[0], caches[1], caches[2])
        assert eng.log == ["import", "step", "export", "step", "step", "step"]
        caches[0][0, 0, 0, 0, 0] = 7.0
        y, caches[0], caches[1], caches[2] = sm5(x, caches[0], caches[1], caches[2])
        assert eng.log[-4:] == ["export", "import", "step", "export"] and float(caches[0][0, 0, 0, 0, 0]) == 8.0
    # (7) hand-over between two models (an A/B, a hot swap via convert_to_stream mid-stream): model B receives the
    #     lazy caches model A handed out with A's export still pending -- A must write them out before B imports
    smA, smB = M.StreamGTCRNMicro().eval(), M.StreamGTCRNMicro().eval()
    eng.log.clear()
    caches = list(smA.init_caches(B, "cpu"))
    for t in range(4):
        y, caches[0], caches[1], caches[2] = smA(x, caches[0], caches[1], caches[2])
    assert eng.log[-1] == "step"                                                      # A owes an export
    y, caches[0], caches[1], caches[2] = smB(x, caches[0], caches[1], caches[2])
    assert eng.log[-4:] == ["export", "import", "step", "export"]
    assert float(caches[0][0, 0, 0, 0, 0]) == 5.0                                     # 4 frames in A + 1 in B
    y, caches[0], caches[1], caches[2] = smA(x, caches[0], caches[1], caches[2])      # and back again
    assert float(caches[0][0, 0, 0, 0, 0]) == 6.0


# ---- rank -> NUMA node binding (sharding.bind_rank_to_gpu_numa) on a fake sysfs tree ---------------------------------

def _fake_sysfs(root, gpus, cpulists):
    """gpus: [(render minor, numa node)] in KFD order behind two CPU nodes; cpulists: {node: "a-b"}."""
    nodes = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    for n in range(2):                                          # CPU nodes: no SIMDs, no render minor
        os.makedirs(os.path.join(nodes, str(n)))
        open(os.path.join(nodes, str(n), "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
    for k, (minor, numa) in enumerate(gpus):
        d = os.path.join(nodes, str(2 + k))
        os.makedirs(d)
        open(os.path.join(d, "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {minor}\n")
        dev = os.path.join(root, f"sys/class/drm/renderD{minor}/device")
        os.makedirs(dev)
        open(os.path.join(dev, "numa_node"), "w").write(f"{numa}\n")
    for node, cl in cpulists.items():
        d = os.path.join(root, f"sys/devices/system/node/node{node}")
        os.makedirs(d)
        open(os.path.join(d, "cpulist"), "w").write(cl + "\n")


def test_rank_is_bound_to_the_numa_node_of_its_gpu(tmp_path, monkeypatch):
    from gtcrn_micro_amd import sharding as S
    assert S.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11] and S.parse_cpulist("") == []
    allowed = sorted(os.sched_getaffinity(0))
    half = max(1, len(allowed) // 2)
    lo, hi = allowed[:half], allowed[half:] or allowed[:half]
    fmt = lambda c: ",".join(str(x) for x in c)
    root = str(tmp_path)
    # eight GPUs, render minors out of order on purpose: four behind each socket
    _fake_sysfs(root, [(128 + k, 0 if k < 4 else 1) for k in (0, 1, 2, 3, 4, 5, 6, 7)], {0: fmt(lo), 1: fmt(hi)})
    assert S.gpu_numa_nodes(root) == [0, 0, 0, 0, 1, 1, 1, 1]
    for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    i0 = S.bind_rank_to_gpu_numa(1, root, apply=False)
    i1 = S.bind_rank_to_gpu_numa(6, root, apply=False)
    assert i0 == {"numa_node": 0, "cpus": len(lo), "bound": True, "physical_gpu": 1}
    assert i1 == {"numa_node": 1, "cpus": len(hi), "bound": True, "physical_gpu": 6}
    # a launcher that remaps the devices: local rank 0 is physical GPU 5 -> node 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5,6")
    assert S.bind_rank_to_gpu_numa(0, root, apply=False)["numa_node"] == 1
    # the lists compose: HIP_VISIBLE_DEVICES indexes INTO what ROCR_VISIBLE_DEVICES leaves visible -- local rank 1 ->
    # HIP entry "0" -> ROCR entry "2" -> physical GPU 2 (node 0); local rank 0 -> "1" -> "7" (node 1)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "2,7")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")
    got = S.bind_rank_to_gpu_numa(1, root, apply=False)
    assert (got["physical_gpu"], got["numa_node"]) == (2, 0)
    got = S.bind_rank_to_gpu_numa(0, root, apply=False)
    assert (got["physical_gpu"], got["numa_node"]) == (7, 1)
    # a list that cannot be resolved (UUID form, an index past its end): no guess, no binding
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "GPU-6f5e4d3c2b1a0000,GPU-0000a1b2c3d4e5f6")
    assert S.bind_rank_to_gpu_numa(0, root, apply=False) == {"numa_node": None, "cpus": None, "bound": False, "physical_gpu": "unknown"}
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "3")
    assert S.bind_rank_to_gpu_numa(0, root, apply=False)["bound"] is False          # HIP entry "1" of a one-entry ROCR list
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    # really applied (and restored): the process mask becomes the node's CPUs
    try:
        got = S.bind_rank_to_gpu_numa(2, root, apply=True)
        assert got["bound"] and sorted(os.sched_getaffinity(0)) == lo
    finally:
        os.sched_setaffinity(0, allowed)
    # nothing to go by: no topology, a GPU without a node, CPUs we may not use -> the mask is left alone
    assert S.bind_rank_to_gpu_numa(0, str(tmp_path / "nowhere")) == {"numa_node": None, "cpus": None, "bound": False, "physical_gpu": 0}
    root2 = str(tmp_path / "single")
    _fake_sysfs(root2, [(128, -1)], {})
    assert S.bind_rank_to_gpu_numa(0, root2)["bound"] is False
    root3 = str(tmp_path / "foreign")
    _fake_sysfs(root3, [(128, 0)], {0: "100000-100003"})
    assert S.bind_rank_to_gpu_numa(0, root3) == {"numa_node": 0, "cpus": None, "bound": False, "physical_gpu": 0}
    assert sorted(os.sched_getaffinity(0)) == allowed


def test_sharding_module_needs_no_torch_at_import():
    """bench.py binds the rank's CPUs BEFORE torch / HIP are imported: sharding.py must import with the stdlib alone."""
    import ast
    src = open(os.path.join(ROOT, "gtcrn_micro_amd", "sharding.py")).read()
    top = [n for n in ast.parse(src).body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert [a.name for n in top for a in n.names] == ["os"]


def test_folder_driver_wav_writer_is_scipys_byte_for_byte(tmp_path):
    """The folder driver writes its 16-bit files with a hand-packed canonical header (infer.write_wav_pcm16; the pipelined
    form packs the same header around int16 samples converted on the device): the bytes on disk are
    scipy.io.wavfile.write's -- lengths 0 and 1, clipping and rounding ties included -- and a mono 16-bit file's samples
    start at the offset pass 1 learns from the memory map."""
    from scipy.io import wavfile
    from gtcrn_micro_amd.infer import read_wav_f32, write_wav_pcm16
    rng = np.random.default_rng(5)
    for k, n in enumerate((0, 1, 257, 12345)):
        x = (rng.standard_normal(n) * 0.6).astype(np.float32)
        if n > 4:
            x[:5] = [1.5, -1.5, 0.5 / 32768.0, 1.5 / 32768.0, -2.5 / 32768.0]      # clipping; rint's ties to even
        a, b = tmp_path / f"a{k}.wav", tmp_path / f"b{k}.wav"
        write_wav_pcm16(str(a), x)
        wavfile.write(str(b), 16000, np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16))
        assert a.read_bytes() == b.read_bytes(), n
        if n:
            fs, m = wavfile.read(str(b), mmap=True)
            raw = np.fromfile(str(b), dtype="<i2", count=n, offset=int(m.offset))
            assert np.array_equal(raw.astype(np.float32) * np.float32(1.0 / 32768.0), read_wav_f32(str(b))[1])
