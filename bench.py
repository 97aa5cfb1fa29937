#!/usr/bin/env python3
"""bench.py -- headline benchmark of the GTCRN-Micro hot path on MI355X.

Metric (BASELINE.json): 16 kHz frames/s through STFT -> mask -> iSTFT.
Workload (BASELINE.json configs[1]): batch of 256 four-second 16 kHz clips per GPU, fp32,
offline forward, wave -> wave (the caller loop of infer.py:60-76), synthetic N(0, 0.1^2) input
already resident in HBM, shipped checkpoint weights (tests/golden/params_dns3.f32).
One "step" = one pass of the fused kernels over the batch = 256 x 251 = 64 256 frames per GPU.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks (one process per GPU,
`python -m torch.distributed.run` on 127.0.0.1, like the reference's own mp.spawn, train.py:461-471) BEFORE
anything in this process touches the GPU, relays rank 0's JSON line and exits non-zero if a rank fails or
the line does not say n_gpus == N.  Launched under an external torchrun it reads RANK/LOCAL_RANK/WORLD_SIZE
and refuses (non-zero exit) a WORLD_SIZE that differs from --gpus.

Multi-GPU: utterances are independent, so every rank runs its own 256-clip shard on its own GPU with no
data-path collective (weak scaling); torch.distributed (RCCL) is used only for the barrier and the
max-over-ranks of the elapsed time.

Besides the headline fields the JSON line carries
  roofline      the dominant kernel: algorithmic fp32 FLOPs per launch / average launch duration from HIP
                events recorded on the launch stream inside the timed region, against the dense fp32 MFMA
                peak (157.3 TFLOP/s, MI355X_MICROARCH.md); `traffic` = HBM bytes per launch from the
                committed rocprofv3 PMC run of this round when profiles/ has it, else null
  cpu_baseline  the PyTorch-CPU port of the path (oracle/torch_port.py, the reference's own ATen op
                sequence) timed on this box's host cores on a bounded sample (rank 0, N = 1 only)
  stream        BASELINE configs[2]: 1024 concurrent streams per GPU, one 16 ms frame per call, state
                resident on the device: per-call latency mean/p50/p99, RTF, frame-steps/s, fraction of the
                94 KB/frame-stream state-traffic bound
  train         BASELINE configs[3]: full train steps at B=512 clips per GPU (STFT x2, train-mode forward,
                HybridLoss, backward, ONE gradient all-reduce over RCCL when N > 1, clip, Adam)
  quant         BASELINE configs[4]: the int8-weight / fp16-activation variant when the library has it
These secondary legs run AFTER the headline timed region and never change the headline fields.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic multiply-accumulates per frame, per kernel (SURVEY.md 2a with the
# ERB filterbank counted sparse as it is computed): 1 MAC = 2 FLOP
MAC_FRONT = 1146 + 1161 + 15600 + 42240                  # ERB.bm, SFE, en_conv0, en_conv1
MAC_GT_ENC = 13288                                       # one encoder GTConvBlock (+TRA)
MAC_PER_FRAME = {
    "k_front": MAC_FRONT,                                # round 2: STFT + front end fused (frames stay in LDS)
    "k_encoder": MAC_FRONT + 3 * MAC_GT_ENC,             # round 1 form (front end inside the encoder)
    "k_encoder_gt": 3 * MAC_GT_ENC,
    "k_gtcn1": 73920,
    "k_gtcn2": 73920,
    "k_decoder": 3 * 84602 + 42240 + 10400 + 764 + 1028,
}
# MFMA instructions (16x16x4 f32 = 2 048 FLOP each) issued per frame, incl. the zero padding of the 8->16 /
# 16->8 pointwise and K=15 tiles: executed vs algorithmic matrix work
MFMA_PER_FRAME = {"k_encoder": 107, "k_gtcn1": 66, "k_gtcn2": 66, "k_decoder": 66.0}
# round 3: the decoder's dense transposed 3x3 and de_convs.3 run on v_mfma_f32_16x16x32_bf16 (16 384 FLOP each) from an
# exact three-way bf16 split of both operands, six partial products per K-chunk: 33/16 tiles x (3 blocks x 30 + 18)
# per frame; the fp32 MFMA keeps the pointwise convs and de_convs.4: 33/16 x (3 x 8 + 8)
BF16_MFMA_PER_FRAME = {"k_decoder": 222.75}
BF16_MFMA_PEAK_TFLOPS = 2500.0                  # MI355X_MICROARCH.md, dense
MODEL_MAC_PER_FRAME = MAC_FRONT + 3 * MAC_GT_ENC + 2 * 73920 + MAC_PER_FRAME["k_decoder"]
FFT_FLOP_PER_FRAME = 2 * 11520 + 6000          # 512-point rFFT + irFFT (5 N log2 N / 2) + window/OLA
FP32_MFMA_PEAK_TFLOPS = 157.3                   # MI355X_MICROARCH.md, dense, v_mfma_f32_16x16x4_f32
HBM_PEAK_GBS = 8000.0
STREAM_STATE_BYTES_PER_FRAME = 94 * 1024        # ring-state traffic + the frame itself
# counter-measured HBM traffic of one B = 512 x 4 s train step per storage mode (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
# passes, FETCH doubled for the 16-byte-per-lane reads; tools/profile_summary.py); a mode without counters reports null
# (round 4: every storage mode, dispatch by dispatch.  The x2 on FETCH_SIZE is calibrated for 16-byte-per-lane reads; the
# 16-bit tensors are read 8 bytes per lane, for which the guide gives no factor: for the two 16-bit modes the figure
# below is the UPPER reading, `traffic_lower` the raw one -- the truth lies between)
ROUND_TAG = "r06"


def train_hbm_bytes():
    """Counter-measured HBM bytes of one B = 512 x 4 s train step per storage mode, read from the newest committed profile
    (tools/profile_summary.py -> profiles/<tag>_train_hbm_traffic.json; FETCH_SIZE x 2 + WRITE_SIZE: the x 2 holds at 4, 8
    and 16 bytes per lane, profiles/r05_fetch_calibration.json -- ONE figure per mode since round 5).  A constant of the
    profiled build, not of the run that prints it: the source file is named in the line."""
    for tag in (ROUND_TAG, "r05", "r04"):
        fn = os.path.join(ROOT, "profiles", f"{tag}_train_hbm_traffic.json")
        try:
            modes = json.load(open(fn)).get("per_storage_mode", {})
        except (OSError, ValueError):
            continue
        out = {}
        for m, v in modes.items():
            gb = v.get("GB_per_step", v.get("GB_per_step_fetch_x2"))
            if gb:
                out[m] = gb * 1e9
        if out:
            return out, f"profiles/{tag}_train_hbm_traffic.json"
    return {}, None


TRAIN_HBM_BYTES_PER_STEP, TRAIN_HBM_SOURCE_FILE = train_hbm_bytes()
WATCHDOG_EXIT_CODE = 3                                  # exit status of every rank when a watchdog had to cut a leg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: start N ranks as children (nothing in this process has imported torch
    or touched the GPU), relay their output, insist on one JSON line with n_gpus == N."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(p.stdout)
    sys.stdout.flush()
    if p.returncode != 0:
        print(f"bench.py: a rank failed (launcher exit code {p.returncode})", file=sys.stderr)
        return p.returncode or 1
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    if len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    if json.loads(lines[0]).get("n_gpus") != args.gpus:
        print(f"bench.py: the line says n_gpus={json.loads(lines[0]).get('n_gpus')} but --gpus {args.gpus}",
              file=sys.stderr)
        return 1
    return 0


def _load_shim():
    """TEST ONLY: tests/test_dist_gloo.py points GTCRN_BENCH_TEST_SHIM at a module that supplies a stand-in
    engine so the multi-rank control flow can run on CPU over gloo.  bench.py itself holds no fake engine."""
    path = os.environ.get("GTCRN_BENCH_TEST_SHIM")
    if not path:
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("gtcrn_bench_test_shim", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class Watchdog:
    """Bounds a phase that may hang on a collective (a rank that died alone leaves the others waiting in RCCL until its
    own timeout, long after the driver has given up): after `seconds` rank 0 prints the line as it stands -- the
    headline is complete before any secondary leg starts -- with the reason, and every rank leaves the process with a
    NON-ZERO status (WATCHDOG_EXIT_CODE): the line is kept, the failure is not hidden from torchrun / the driver."""

    def __init__(self, seconds, rank, get_line, what):
        self.seconds, self.rank, self.get_line, self.what = seconds, rank, get_line, what
        self.timer = None

    def _fire(self):
        if self.rank == 0:
            line = self.get_line()                             # None: the line has been printed already
            if line is not None:
                line.setdefault("secondary_errors", {})[self.what] = (
                    f"timed out after {self.seconds:g} s (a rank failed or hung inside a collective); line printed by "
                    "the watchdog")
                print(json.dumps({k: v for k, v in line.items() if not k.startswith("_")}), flush=True)
        else:
            time.sleep(2.0)                                    # let rank 0's line out first
        sys.stdout.flush()
        # a rank hung or died inside a collective: the line (with the reason) is out, but the run did NOT succeed --
        # the launcher and the driver must see a failure.  No re-exec, no clean-up through the hung runtime.
        os._exit(WATCHDOG_EXIT_CODE)

    def __enter__(self):
        import threading
        self.timer = threading.Timer(self.seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def gather_ranks(obj, world):
    """[obj of rank 0, ..., obj of rank world-1] on every rank (identity when not distributed)."""
    if world == 1:
        return [obj]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def run_leg(name, fn, rank, world, get_line, timeout_s=300.0):
    """One secondary leg, tolerant of a rank failing alone: `fn(sync_local, record_elapsed)` runs WITHOUT collectives
    (every rank times its own shard between device synchronisations; the ranks start together behind one barrier
    OUTSIDE the try block), then ONE object gather, also outside the try block, carries every rank's result or error
    to rank 0, which prices throughput with the slowest rank's time.  Legs whose work itself contains a collective
    (the train step's gradient all-reduce) validate their allocations locally first and agree on it (see
    train_leg).  The whole leg sits inside a watchdog."""
    import torch
    els = []

    def sync_local():
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def record(v, device=None):
        els.append(float(v))
        return float(v)

    with Watchdog(timeout_s, rank, get_line, name):
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        res, err = None, None
        try:
            res = fn(sync_local, record)
        except Exception as e:                                 # a secondary leg must never take the headline down
            err = repr(e)
        allr = gather_ranks({"res": res if rank == 0 else None, "err": err, "el": els[-1] if els else None}, world)
    errs = {str(r): x["err"] for r, x in enumerate(allr) if x["err"]}
    if errs:
        return {"error": errs["0"] if world == 1 else errs}
    res = allr[0]["res"]
    if res is None:
        return None
    tl = [x["el"] or 0.0 for x in allr]
    k = tl[0] / max(tl) if max(tl) > 0 else 1.0                # rank 0 priced its own time; the slowest rank sets the rate
    for key in res.pop("_rate_keys", []):
        res[key] = round(res[key] * k, 1)
    for key in res.pop("_time_keys", []):
        res[key] = round(res[key] / k, 4)
    if world > 1:
        res["per_rank_timed_s"] = {"min": round(min(tl), 5), "max": round(max(tl), 5)}
    return res


def cpu_model_name():
    """The host CPU as /proc/cpuinfo names it ("state the core count and CPU model")."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


_CPU_HOST_WORKER = r"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
from oracle.torch_port import TorchPort
threads, seconds = int(sys.argv[2]), float(sys.argv[3])
torch.set_num_threads(threads)
port = TorchPort(np.fromfile(os.path.join(sys.argv[1], "tests", "golden", "params_dns3.f32"), dtype=np.float32))
win = torch.hann_window(512).pow(0.5)
x = torch.randn(16, 64000, generator=torch.Generator().manual_seed(43 + os.getpid() % 97)) * 0.1
port.enhance(x, win)
n, t0 = 0, time.perf_counter()
while True:
    port.enhance(x, win)
    n += 1
    el = time.perf_counter() - t0
    if el >= seconds:
        break
print("RATE", n * 16 * 251 / el)
"""


def cpu_host_wide(threads_each, seconds=5.0, max_procs=16, limit_s=180.0):
    """The same CPU port as several concurrent processes (each `threads_each` threads, its own B = 16 batches): what the
    HOST -- not one PyTorch process, which stops scaling at 16-32 threads on this 19 k-parameter model -- gets through.
    Plain child processes running a few lines of Python (no GPU context, nothing shared with this process), ~5 s each,
    all at once, under a time limit."""
    import subprocess
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(max_procs, ncpu // max(1, threads_each)))
    if procs < 2:
        return None
    env = dict(os.environ, OMP_NUM_THREADS=str(threads_each), MKL_NUM_THREADS=str(threads_each))
    kids = [subprocess.Popen([sys.executable, "-c", _CPU_HOST_WORKER, ROOT, str(threads_each), str(seconds)],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(procs)]
    rates, t_end = [], time.time() + limit_s
    for k in kids:
        try:
            out, _ = k.communicate(timeout=max(1.0, t_end - time.time()))
            rates += [float(l.split()[1]) for l in out.splitlines() if l.startswith("RATE")]
        except subprocess.TimeoutExpired:
            k.kill()
            k.communicate()
    if len(rates) < procs:
        return {"error": f"{procs - len(rates)} of {procs} CPU processes did not finish within {limit_s:g} s"}
    return {"value": round(sum(rates), 1), "unit": "frames/s", "processes": procs, "threads_each": int(threads_each),
            "cores": int(procs * threads_each),
            "sample": f"{procs} concurrent processes x {threads_each} threads, each passing B=16 four-second clips through "
                      f"oracle/torch_port.py for {seconds:g} s; the sum of their rates"}


def cpu_baseline(params, seconds_per_candidate=3.0):
    """The reference's CPU arithmetic (ATen) on a bounded sample of the same workload.

    The thread count matters a lot for this 19 k-parameter model (128 threads on a 256-core host are slower
    than 16), so every candidate gets a >= 3 s run at B=16 and the best one is reported; the utterance-at-a-time
    shape of infer.py (B=1) is swept and reported separately."""
    import torch
    from oracle.torch_port import TorchPort
    port = TorchPort(params)
    win = torch.hann_window(512).pow(0.5)
    g = torch.Generator().manual_seed(43)
    x = torch.randn(16, 64000, generator=g) * 0.1
    default_threads = torch.get_num_threads()

    def rate(seconds, xin):
        port.enhance(xin, win)                             # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            port.enhance(xin, win)
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds:
                return n * xin.shape[0] * 251 / el, n, el

    cands = [th for th in (1, 8, 16, 32) if th <= (os.cpu_count() or 1)]
    sweep16, sweep1, passes = {}, {}, {}
    for th in cands:
        torch.set_num_threads(th)
        sweep16[th], n, el = rate(seconds_per_candidate, x)
        passes[th] = (n, el)
        sweep1[th] = rate(seconds_per_candidate / 2, x[:1])[0]
    torch.set_num_threads(default_threads)
    best16 = max(sweep16, key=sweep16.get)
    best1 = max(sweep1, key=sweep1.get)
    n, el = passes[best16]
    # one process stops scaling at 16-32 threads; the host has more cores: the same port as concurrent processes
    try:
        host = cpu_host_wide(min(best16, 16))
    except Exception as e:                                    # (a host that cannot spawn: the single-process figure stands)
        host = {"error": repr(e)}
    return {
        "host_wide": host,
        "value": round(sweep16[best16], 1), "unit": "frames/s", "cores": int(best16), "kind": "port",
        "sample": f"{n} passes of 16 four-second clips (B=16, {el:.1f} s) through oracle/torch_port.py "
                  f"(PyTorch {torch.__version__} CPU, the reference's ATen op sequence) at {best16} threads, the "
                  f"best of {cands} (>= {seconds_per_candidate:g} s each)",
        "b1_infer_py_style": {"value": round(sweep1[best1], 1), "cores": int(best1),
                              "sample": "one four-second clip per call (infer.py:48-107)"},
        "thread_sweep_b16_fps": {str(k): round(v) for k, v in sweep16.items()},
        "thread_sweep_b1_fps": {str(k): round(v) for k, v in sweep1.items()},
        "host_cpus": os.cpu_count(),
        "cpu_model": cpu_model_name(),
    }


def stream_leg(eng, world, sync_all, max_over_ranks, nstreams=1024, frames=251):
    """BASELINE configs[2]: `nstreams` concurrent streams per GPU, one frame per call (the loop of
    gtcrn_micro_stream.py:626-635 with the state left on the device)."""
    import numpy as np
    import torch
    N, T = nstreams, frames
    torch.manual_seed(44)
    spec = (torch.randn(N, T, 257, 2, device="cuda") * 0.3).permute(0, 2, 1, 3)   # frame-major storage
    out = torch.empty((N, T, 257, 2), device="cuda").permute(0, 2, 1, 3)
    eng.reserve(N, 1)
    state = eng.new_state(N)
    for t in range(8):
        eng.stream_step(state, spec[:, :, t:t + 1], out=out[:, :, t:t + 1])
    state = eng.new_state(N)
    sync_all()
    lat = []
    for t in range(T):                                            # latency: one call, wait for it
        t0 = time.perf_counter()
        eng.stream_step(state, spec[:, :, t:t + 1], out=out[:, :, t:t + 1])
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e3
    full = eng.forward_spec(spec.contiguous())
    err = float((full - out).abs().max() / full.abs().max())
    state = eng.new_state(N)
    sync_all()
    t0 = time.perf_counter()
    for t in range(T):                                            # throughput: calls queued back to back
        eng.stream_step(state, spec[:, :, t:t + 1], out=out[:, :, t:t + 1])
    sync_all()
    el = max_over_ranks(time.perf_counter() - t0, "cuda")
    fsteps = world * N * T / el
    extra = stream_extras(eng, spec, N)
    res = {
        "workload": f"{N} concurrent streams per GPU x {T} single-frame calls (hop 256), state resident in HBM",
        "streams_per_gpu": N, "calls": T,
        "latency_ms": {"mean": round(float(lat.mean()), 4), "p50": round(float(np.percentile(lat, 50)), 4),
                       "p99": round(float(np.percentile(lat, 99)), 4), "max": round(float(lat.max()), 4)},
        "rtf_per_stream": round(float(lat.mean()) / 16.0, 6),
        "frame_steps_per_s": round(fsteps, 1),
        "ms_per_call_back_to_back": round(el / T * 1e3, 4),
        "state_bound_frame_steps_per_s": round(HBM_PEAK_GBS * 1e9 / STREAM_STATE_BYTES_PER_FRAME, 1),
        "frac_of_state_bound": round(fsteps / world / (HBM_PEAK_GBS * 1e9 / STREAM_STATE_BYTES_PER_FRAME), 4),
        "bound_note": f"at N = {N} the state ({N * eng.state_bytes() / 2**20:.0f} MiB) is resident in the 256 MiB Infinity "
                      "Cache: the 94 KB/frame-stream HBM bound is NOT what limits this step (the serial chain of ~35 "
                      "barrier phases per workgroup does, profiles/r04_phase_profile_stream.txt); see stream_capacity "
                      "for the sizes where the bound is real",
        "stream_vs_offline_rel_err": err, "dtype": "f32",
        "launches_per_call": 1,
        **extra,
        "_rate_keys": ["frame_steps_per_s"], "_time_keys": ["ms_per_call_back_to_back"],
    }
    del spec, out, state, full
    torch.cuda.empty_cache()
    return res


def stream_extras(eng, spec, N, frames=200):
    """Rank-local companions of the stream numbers (no collective): the step replayed from a captured HIP graph, ONE stream (the reference's own loop, gtcrn_micro_stream.py:618-635) and the
    reference-shaped call `StreamGTCRNMicro.forward` (caller-owned caches handed back every frame) next to the native
    `step`, for one stream and for N."""
    import numpy as np
    import torch

    def b2b(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(n):
            fn(t)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def lat(fn, n):
        v = []
        for t in range(n):
            t0 = time.perf_counter()
            fn(t)
            torch.cuda.synchronize()
            v.append(time.perf_counter() - t0)
        v = np.array(v[5:]) * 1e3
        return {"mean": round(float(v.mean()), 4), "p50": round(float(np.percentile(v, 50)), 4),
                "p99": round(float(np.percentile(v, 99)), 4)}

    T = spec.shape[2]
    out = {}
    y = torch.empty((N, 1, 257, 2), device="cuda").permute(0, 2, 1, 3)
    # (b) the step replayed from a HIP graph
    try:
        x = torch.empty((N, 1, 257, 2), device="cuda").permute(0, 2, 1, 3)
        stg = eng.new_state(N)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            x.copy_(spec[:, :, :1])
            eng.stream_step(stg, x, out=y)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                eng.stream_step(stg, x, out=y)
        out["graph_replay_ms_per_call_back_to_back"] = round(b2b(lambda t: g.replay(), frames), 4)
        del g, stg
    except Exception as e:
        out["graph_replay_ms_per_call_back_to_back"] = repr(e)
    # (c) ONE stream, native step
    st1 = eng.new_state(1)
    s1 = spec[:1]
    y1 = torch.empty((1, 1, 257, 2), device="cuda").permute(0, 2, 1, 3)
    out["one_stream"] = {"latency_ms": lat(lambda t: eng.stream_step(st1, s1[:, :, t % T:t % T + 1], out=y1), frames),
                         "ms_per_call_back_to_back": round(b2b(lambda t: eng.stream_step(st1, s1[:, :, t % T:t % T + 1], out=y1), frames), 4)}
    out["one_stream"]["rtf"] = round(out["one_stream"]["latency_ms"]["mean"] / 16.0, 6)
    # (d) the reference-shaped call with caller-owned caches
    try:
        from gtcrn_micro_amd.streaming.gtcrn_micro_stream import StreamGTCRNMicro
        from gtcrn_micro_amd.models.gtcrn_micro import load_blob_into
        sm = StreamGTCRNMicro().eval()
        load_blob_into(sm, np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32))
        sm = sm.to("cuda")
        mirror = {}
        with torch.no_grad():
            for nn_, sp in ((1, spec[:1]), (N, spec)):
                caches = list(sm.init_caches(nn_, "cuda"))

                def call(t, sp=sp, caches=caches):
                    _, caches[0], caches[1], caches[2] = sm(sp[:, :, t % T:t % T + 1], caches[0], caches[1], caches[2])
                for t in range(5):
                    call(t)
                stn = eng.new_state(nn_)
                mirror[f"n{nn_}"] = {
                    "forward_ms_per_call_back_to_back": round(b2b(call, frames), 4),
                    "native_step_ms_per_call_back_to_back": round(b2b(lambda t, sp=sp, stn=stn: eng.stream_step(stn, sp[:, :, t % T:t % T + 1]), frames), 4)}
                if nn_ == 1:
                    mirror["n1"]["forward_latency_ms"] = lat(call, frames)
            mirror["imports"], mirror["exports"], mirror["calls"] = (sm.forward_stats[k] for k in ("imports", "exports", "calls"))
        out["reference_shaped_forward"] = mirror
    except Exception as e:
        out["reference_shaped_forward"] = {"error": repr(e)}
    return out


def io_leg(eng, wave, win, out, world, sync_all, max_over_ranks, steps=20):
    """The host link of the caller loop ("exclude H2D of the input (report separately)"; train.py:246,255
    `noisy.to(device)`, infer.py:60-71 feeds host arrays).  The headline keeps the batch resident; a caller that owns
    HOST buffers pays the link both ways: B x L x 4 bytes in and out per step.  Reported: pinned H2D / D2H rates for the
    batch, the naive serial form (pageable .to() -> kernels -> .cpu()), and the SERVED rate -- three HIP streams, double
    buffered pinned staging, copy-in || six kernels || copy-out -- which is what the link lets through."""
    import torch
    B, L = wave.shape
    nbytes = wave.numel() * 4
    dev = wave.device
    hin = [torch.empty(wave.shape, dtype=torch.float32, pin_memory=True) for _ in range(2)]
    hout = [torch.empty(out.shape, dtype=torch.float32, pin_memory=True) for _ in range(2)]
    din = [torch.empty_like(wave) for _ in range(2)]
    dout = [torch.empty_like(out) for _ in range(2)]
    for h in hin:
        h.copy_(wave)
    torch.cuda.synchronize()

    def rate(fn, n=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return nbytes * n / (time.perf_counter() - t0) / 1e9

    h2d = rate(lambda: din[0].copy_(hin[0], non_blocking=True))
    d2h = rate(lambda: hout[0].copy_(dout[0], non_blocking=True))
    # both directions at once (the served pipeline's steady state): two streams
    s_in, s_cmp, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()

    def duplex():
        with torch.cuda.stream(s_in):
            din[0].copy_(hin[0], non_blocking=True)
        with torch.cuda.stream(s_out):
            hout[0].copy_(dout[0], non_blocking=True)
    duplex_cold = rate(duplex)
    for _ in range(3):                                # the runtime needs a few dozen concurrent pairs before it drives the
        rate(duplex)                                  # two directions on separate engines (tools/io_overlap_probe.py)
    duplex_each = rate(duplex)
    # naive serial caller: pageable host tensors, synchronous copies
    hp = wave.cpu()
    t0 = time.perf_counter()
    for _ in range(3):
        y = eng.forward_wave(hp.to(dev), win).cpu()
    serial_s = (time.perf_counter() - t0) / 3
    del y
    # served: copy-in || kernels || copy-out over two staging slots
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_cmp = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]

    # The same pipeline with 16-bit PCM at the boundary (what the reference's data are: infer.py:54 reads 16-bit WAV files,
    # :113 writes them): int16 samples in, int16 samples out, widened / rounded on the device (gtcrn_pcm16_to_f32 /
    # gtcrn_f32_to_pcm16, two extra passes of ~0.03 ms) -- HALF the bytes over the link, which is what bounds the float32
    # form.  (Round 5's other shape -- one direction copied by a kernel instead of a DMA engine -- lost inside the pipeline,
    # profiles/r05_link_copy_probe.txt, and left the ABI.)
    from gtcrn_micro_amd import f32_to_pcm16, pcm16_to_f32
    pcm = [False]
    hin16 = [torch.empty(wave.shape, dtype=torch.int16, pin_memory=True) for _ in range(2)]
    hout16 = [torch.empty(out.shape, dtype=torch.int16, pin_memory=True) for _ in range(2)]
    din16 = [torch.empty(wave.shape, dtype=torch.int16, device=dev) for _ in range(2)]
    dout16 = [torch.empty(out.shape, dtype=torch.int16, device=dev) for _ in range(2)]
    wave16 = f32_to_pcm16(wave.contiguous())
    for h in hin16:
        h.copy_(wave16)
    torch.cuda.synchronize()

    def served(n):
        for i in range(n):
            k = i & 1
            with torch.cuda.stream(s_in):
                s_in.wait_event(ev_cmp[k])            # slot k's previous batch has been consumed by the kernels
                if pcm[0]:
                    din16[k].copy_(hin16[k], non_blocking=True)
                else:
                    din[k].copy_(hin[k], non_blocking=True)
                ev_in[k].record()
            with torch.cuda.stream(s_cmp):
                s_cmp.wait_event(ev_in[k])
                s_cmp.wait_event(ev_out[k])           # ... and its previous output has left the device
                if pcm[0]:
                    pcm16_to_f32(din16[k], out=din[k])
                eng.forward_wave(din[k], win, out=dout[k])
                if pcm[0]:
                    f32_to_pcm16(dout[k], out=dout16[k])
                ev_cmp[k].record()
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_cmp[k])
                if pcm[0]:
                    hout16[k].copy_(dout16[k], non_blocking=True)
                else:
                    hout[k].copy_(dout[k], non_blocking=True)
                ev_out[k].record()

    def steady():
        served(4)
        sync_all()
        t0 = time.perf_counter()
        served(steps)
        sync_all()
        cold = time.perf_counter() - t0               # the first region after the pipeline starts (transient)
        served(3 * steps)
        regions = []
        for _ in range(3):
            sync_all()
            t0 = time.perf_counter()
            served(steps)
            sync_all()
            regions.append(time.perf_counter() - t0)
        # steady state: the median of three regions of `steps` steps
        return cold, max_over_ranks(sorted(regions)[1], "cuda")
    cold_el, el = steady()
    same = bool(torch.equal(hout[(steps - 1) & 1], out.cpu()))     # the served output is the resident path's, bit for bit
    pcm[0] = True
    _, el_pcm = steady()
    # the PCM16 pipeline's output is the resident path's on the int16-valued input, rounded once: checked against the
    # resident kernels fed with the widened samples and torch's own rounding
    ref16 = torch.clamp(torch.round(eng.forward_wave(wave16.to(torch.float32) / 32768.0, win) * 32768.0), -32768, 32767).to(torch.int16)
    same16 = bool(torch.equal(hout16[(steps - 1) & 1], ref16.cpu()))
    pcm[0] = False
    T = 1 + L // 256
    res = {
        "workload": f"the headline batch handed over as HOST buffers: {nbytes / 1e6:.1f} MB in + "
                    f"{out.numel() * 4 / 1e6:.1f} MB out per step over the host link",
        "h2d_GBps": round(h2d, 2), "d2h_GBps": round(d2h, 2), "duplex_each_GBps": round(duplex_each, 2),
        "duplex_each_GBps_first_pairs": round(duplex_cold, 2),
        "served_ms_per_step_first_region": round(cold_el / steps * 1e3, 4),
        "pinned": True,
        "served_frames_per_s": round(world * B * T * steps / el, 1),
        "served_ms_per_step": round(el / steps * 1e3, 4),
        "served_pipeline": "3 HIP streams, 2 pinned staging slots: copy-in || 6 kernels || copy-out (both copies on DMA engines)",
        "served_pcm16_frames_per_s": round(world * B * T * steps / el_pcm, 1),
        "served_pcm16_ms_per_step": round(el_pcm / steps * 1e3, 4),
        "served_pcm16_pipeline": "the same three streams with int16 samples over the link both ways (half the bytes), "
                                 "int16 <-> float32 on the device (gtcrn_pcm16_to_f32 / gtcrn_f32_to_pcm16)",
        "served_equals_resident": same, "served_pcm16_equals_resident_rounded": same16,
        "served_pcm16_note": "half the link bytes, yet not faster than the float32 pipeline on this platform: the two conversion "
                             "launches cost 0.02 ms each alone, but with both directions busy the int16 copies run at the "
                             "half duplex rate (24-28 GB/s each way: 1.2-1.4 ms for 32.8 MB, tools/pcm_probe.py), so the "
                             "pipeline stays copy bound; the format pays where the HOST is the bound (the folder driver: "
                             "no per-sample host work)",
        "serial_pageable_frames_per_s": round(B * T / serial_s, 1),
        "serial_pageable_ms_per_step": round(serial_s * 1e3, 3),
        "link_bound_frames_per_s": round(min(h2d, d2h, duplex_each) * 1e9 / (L * 4) * T, 1),
        "note": "the headline `value` is the resident rate (inputs in HBM when the timed region starts); "
                "served_frames_per_s is what a caller with host buffers gets in steady state and is bounded by the "
                "link (link_bound_frames_per_s = the slowest of h2d, d2h and the per-direction rate with both "
                "directions busy / bytes per clip x frames per clip), not by the kernels; the first few dozen steps of a "
                "pipeline run at about half the duplex rate (the *_first_* fields)",
        "_rate_keys": ["served_frames_per_s", "served_pcm16_frames_per_s"],
        "_time_keys": ["served_ms_per_step", "served_pcm16_ms_per_step"],
    }
    del hin, hout, din, dout, hin16, hout16, din16, dout16
    torch.cuda.empty_cache()
    return res


def folder_leg(rank, world, local_rank, nclips=512):
    """The bulk offline driver (gtcrn_micro_amd/infer.py, counterpart of infer.py:26-119) on a generated folder in
    tmpfs: `nclips` 16-bit clips of 2-10 s.  Pipelined (reader thread, pinned staging, three streams, writer thread)
    against the serial form on the same files; frames/s includes WAV decode / encode on the host."""
    import shutil
    import tempfile
    import numpy as np
    from scipy.io import wavfile
    from gtcrn_micro_amd.infer import enhance_folder
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    root = tempfile.mkdtemp(prefix=f"gtcrn_folder_r{rank}_", dir=base)
    try:
        noisy, clean = os.path.join(root, "noisy"), os.path.join(root, "clean")
        os.makedirs(noisy)
        os.makedirs(clean)
        rng = np.random.default_rng(47 + rank)
        lens = rng.integers(32000, 160001, nclips)
        for k, L in enumerate(lens):
            x = np.clip(rng.standard_normal(int(L)) * 3000, -32768, 32767).astype(np.int16)
            wavfile.write(os.path.join(noisy, f"mix_fileid_{k}.wav"), 16000, x)
            # the references only lend their lengths (header-only reads): sparse files of the right size
            wavfile.write(os.path.join(clean, f"clean_fileid_{k}.wav"), 16000, np.zeros(int(L), np.int16))
        ck = os.path.join(ROOT, "tests", "golden", "params_dns3.f32")
        # Each form ONCE untimed (cold engine creation, page cache, pinned-allocation pools, thread start-up: round 5's
        # single shot ran the pipeline first and charged it all of that), then REPS alternating repetitions of each; the
        # reported figures are medians, the spread is reported with them.
        REPS = 3
        runs = {"pipelined": [], "serial": []}
        for rep in range(REPS + 1):
            for key, pipe in (("pipelined", True), ("serial", False)):
                st = {}
                enh = os.path.join(root, "enh_" + key)
                shutil.rmtree(enh, ignore_errors=True)
                enhance_folder(noisy, clean, enh, ck, device=local_rank, max_batch=64, pipeline=pipe, stats=st)
                if rep:
                    runs[key].append(dict(st))
        out = {}
        for key, rr in runs.items():
            rr = sorted(rr, key=lambda r: r["wall_s"])
            st = rr[len(rr) // 2]
            out[key] = {"wall_s": round(st["wall_s"], 3), "frames_per_s": round(st["frames_per_s"], 1),
                        "setup_s": round(st.get("setup_s", 0.0), 3),
                        "wall_s_min_max": [round(rr[0]["wall_s"], 3), round(rr[-1]["wall_s"], 3)],
                        "gpu_busy_frac": None if st["gpu_busy_frac"] is None else round(st["gpu_busy_frac"], 4)}
        # same bytes on disk from both forms
        same = all(open(os.path.join(root, "enh_pipelined", f), "rb").read() ==
                   open(os.path.join(root, "enh_serial", f), "rb").read()
                   for f in sorted(os.listdir(os.path.join(root, "enh_serial"))) if f.endswith(".wav"))
        return {
            "workload": f"{nclips} wav files of 2-10 s (16-bit, 16 kHz) in tmpfs -> {nclips} enhanced files, batches of 64 "
                        "by length, one GPU",
            "clips": int(nclips), "frames": int(st["frames"]), "audio_s": round(float(lens.sum()) / 16000.0, 1),
            "frames_per_s": out["pipelined"]["frames_per_s"], "wall_s": out["pipelined"]["wall_s"],
            "setup_s": out["pipelined"]["setup_s"],
            "frames_per_s_after_setup": round(st["frames"] / max(out["pipelined"]["wall_s"] - out["pipelined"]["setup_s"], 1e-6), 1),
            "gpu_busy_frac": out["pipelined"]["gpu_busy_frac"],
            "serial": out["serial"], "speedup_over_serial": round(out["serial"]["wall_s"] / out["pipelined"]["wall_s"], 2),
            "files_identical_to_serial": bool(same), "host_cpus": os.cpu_count(),
            "method": f"each form once untimed, then {REPS} alternating repetitions each; medians, wall_s_min_max = the spread",
            "note": "frames/s of the whole driver: WAV decode, pinned staging, H2D, kernels, D2H, length match, 16-bit "
                    "encode and file writes; the kernels are a few percent of it (gpu_busy_frac) -- the driver is bound "
                    "by the host's WAV work, which the pipeline spreads over reader / writer threads",
        }
    finally:
        shutil.rmtree(root, ignore_errors=True)


def stream_capacity_leg(eng, world, sync_all, max_over_ranks, sizes=(1024, 4096, 16384, 65536, 262144)):
    """configs[2] beyond 1024 streams: how many concurrent streams ONE GPU carries in real time.  Per N: ms per
    single-frame step (calls queued back to back), frame-steps/s, the state in GB (152 KB per stream: past 256 MiB it no
    longer fits the Infinity Cache and the 94 KB/frame-stream state-traffic bound of SURVEY.md 8d becomes a real HBM
    bound), the fraction of that bound, and max_realtime_streams = the largest N whose step stays under the 16 ms hop."""
    import torch
    frames = 8
    res = {}
    sb = eng.state_bytes()
    bound = HBM_PEAK_GBS * 1e9 / STREAM_STATE_BYTES_PER_FRAME
    last_el = 0.0
    for N in sizes:
        free, _ = torch.cuda.mem_get_info()
        need = N * (sb + 2 * frames * 2056 + 2056)
        if need > 0.8 * free:
            res[str(N)] = {"skipped": f"needs {need / 2**30:.1f} GiB, {free / 2**30:.1f} GiB free"}
            continue
        torch.manual_seed(48)
        spec = (torch.randn(N, frames, 257, 2, device="cuda") * 0.3).permute(0, 2, 1, 3)
        out = torch.empty((N, 1, 257, 2), device="cuda").permute(0, 2, 1, 3)
        state = eng.new_state(N)
        calls = 200 if N <= 4096 else (64 if N <= 65536 else 24)
        for t in range(4):
            eng.stream_step(state, spec[:, :, t % frames:t % frames + 1], out=out)
        sync_all()
        t0 = time.perf_counter()
        for t in range(calls):
            eng.stream_step(state, spec[:, :, t % frames:t % frames + 1], out=out)
        sync_all()
        last_el = time.perf_counter() - t0
        ms = last_el / calls * 1e3
        # which kernel the library picked for this stream count (k_stream_ms: four streams per workgroup, k_stream_wide: seven)
        kernel = None
        if hasattr(eng, "timing_enable"):
            eng.timing_enable(True)
            eng.stream_step(state, spec[:, :, :1], out=out)
            sync_all()
            kernel = "+".join(eng.timing_read())
            eng.timing_enable(False)

        def timed_form(form, ncalls):
            # the same steps in another form of the step (gtcrn_stream_form): the A/Bs of the default
            try:
                eng.reserve(N, 1)
                eng.stream_form(form)
                for t in range(2):
                    eng.stream_step(state, spec[:, :, t % frames:t % frames + 1], out=out)
                sync_all()
                t0 = time.perf_counter()
                for t in range(ncalls):
                    eng.stream_step(state, spec[:, :, t % frames:t % frames + 1], out=out)
                sync_all()
                return (time.perf_counter() - t0) / ncalls * 1e3
            except Exception:      # a library without that form
                return None
            finally:
                eng.stream_form(0)
        # 1: the three-launch form (encoder / both GTCN stacks / decoder as kernels of their own, hand-offs through HBM);
        # 2 / 3: the one-launch step pinned to four / seven streams per workgroup
        ms3 = ms4 = ms7 = None
        if hasattr(eng, "stream_form"):
            if N <= 65536:
                ms3 = timed_form(1, calls)
            ms4 = timed_form(2, max(8, calls // 2))
            ms7 = timed_form(3, max(8, calls // 2))
        fs = N / (ms * 1e-3)
        res[str(N)] = {"ms_per_step": round(ms, 4), "frame_steps_per_s": round(world * fs, 1), "kernel": kernel,
                       "three_launch_form_ms_per_step": None if ms3 is None else round(ms3, 4),
                       "four_streams_per_workgroup_ms_per_step": None if ms4 is None else round(ms4, 4),
                       "seven_streams_per_workgroup_ms_per_step": None if ms7 is None else round(ms7, 4),
                       "state_GB": round(N * sb / 1e9, 3), "state_in_infinity_cache": bool(N * sb <= 256 * 2**20),
                       "frac_of_state_bound": round(fs / bound, 4),
                       "state_bound_GBps_equiv": round(fs * STREAM_STATE_BYTES_PER_FRAME / 1e9, 1),
                       "rtf_per_stream": round(ms / 16.0, 6), "realtime": bool(ms < 16.0)}
        del spec, out, state
        torch.cuda.empty_cache()
    rt = [int(n) for n, v in res.items() if v.get("realtime")]
    # between the last two measured sizes the step time is linear in N (the per-stream cost is constant past the cache)
    done = sorted((int(n), v["ms_per_step"]) for n, v in res.items() if "ms_per_step" in v)
    est = None
    if len(done) >= 2:
        (n0, m0), (n1, m1) = done[-2], done[-1]
        slope = (m1 - m0) / (n1 - n0)
        if slope > 0:
            est = int(n1 + (16.0 - m1) / slope)
    max_over_ranks(last_el, "cuda")
    return {"workload": "single-frame streaming steps at N concurrent streams per GPU, state resident in HBM",
            "sizes": res, "max_realtime_streams_measured": max(rt) if rt else None,
            "max_realtime_streams_linear_estimate": est,
            "note": "frac_of_state_bound prices 94 KB of ring-state traffic per frame-step against 8 TB/s; it is the "
                    "operative bound only where state_in_infinity_cache is false"}


def batch_sweep_leg(eng, win, world, sync_all, max_over_ranks):
    """Shapes other than the headline's, wave -> wave, per GPU: the reference's own call shape (ONE clip per call,
    infer.py:48: a 31-second example-length clip and a 4-second one), a small batch, the headline batch, one clip more
    than a round of 256 workgroups, and two rounds.  frames/s per shape; `rel` = per-frame cost relative to B = 256."""
    import torch
    shapes = [("1x31s", 1, 496000), ("1x4s", 1, 64000), ("32x4s", 32, 64000), ("256x4s", 256, 64000),
              ("257x4s", 257, 64000), ("512x4s", 512, 64000)]
    res, last = {}, 0.0
    for name, B, L in shapes:
        T = 1 + L // 256
        torch.manual_seed(45)
        x = torch.randn(B, L, device="cuda") * 0.1
        y = torch.empty((B, 256 * (T - 1)), device="cuda")
        eng.reserve(B, T)
        iters = 30 if B * T < 40000 else 12
        for _ in range(3):
            eng.forward_wave(x, win, out=y)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(iters):
            eng.forward_wave(x, win, out=y)
        sync_all()
        last = time.perf_counter() - t0
        el = last / iters
        res[name] = {"B": B, "T": T, "ms_per_call": round(el * 1e3, 4), "frames_per_s": round(world * B * T / el, 1)}
        if B <= 32:
            # the small shapes are launch-bound (six launches + span bookkeeping on ~0.03 ms of work): the same call captured
            # ONCE into a HIP graph and replayed -- what a serving loop with fixed shapes would do (the call is capturable:
            # no allocation, no synchronisation inside; tests/test_gpu_infer.py::test_forward_wave_is_graph_capturable)
            try:
                cs = torch.cuda.Stream()
                cs.wait_stream(torch.cuda.current_stream())
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.stream(cs):
                    eng.forward_wave(x, win, out=y)
                    torch.cuda.synchronize()
                    with torch.cuda.graph(gr, stream=cs):
                        eng.forward_wave(x, win, out=y)
                for _ in range(3):
                    gr.replay()
                sync_all()
                t0 = time.perf_counter()
                for _ in range(iters):
                    gr.replay()
                sync_all()
                ge = (time.perf_counter() - t0) / iters
                # latency of ONE call (enqueue + wait), direct and replayed: what a caller that needs the result sees
                def one(fn, n=20):
                    v = []
                    for _ in range(n):
                        t1 = time.perf_counter()
                        fn()
                        torch.cuda.synchronize()
                        v.append(time.perf_counter() - t1)
                    return sorted(v)[n // 2]
                res[name]["graph_replay"] = {"ms_per_call": round(ge * 1e3, 4),
                                             "frames_per_s": round(world * B * T / ge, 1),
                                             "latency_ms_direct": round(one(lambda: eng.forward_wave(x, win, out=y)) * 1e3, 4),
                                             "latency_ms_replay": round(one(gr.replay) * 1e3, 4)}
                del gr
            except Exception as e:
                res[name]["graph_replay"] = {"error": repr(e)}
        del x, y
    base = res["256x4s"]["ms_per_call"] / (256 * 251)
    for v in res.values():
        v["per_frame_cost_rel_256"] = round(v["ms_per_call"] / (v["B"] * v["T"]) / base, 3)
        if "ms_per_call" in v.get("graph_replay", {}):
            v["graph_replay"]["per_frame_cost_rel_256"] = round(v["graph_replay"]["ms_per_call"] / (v["B"] * v["T"]) / base, 3)
    # a folder-like batch (enhance_folder's call): 48 clips of 2 .. 10 s in one variable-length launch sequence, with the
    # per-utterance kernels in time spans over the lengths (the default) and with one workgroup per utterance
    B, L = 48, 160000
    lens = torch.randint(32000, L + 1, (B,), generator=torch.Generator().manual_seed(46))
    frames = int((1 + lens // 256).sum())
    x = torch.randn(B, L, device="cuda") * 0.1
    y = torch.empty((B, 256 * (L // 256)), device="cuda")
    eng.reserve(B, 1 + L // 256)
    var = {"B": B, "frames": frames}
    try:
        for key, on in (("spans", True), ("one_workgroup_per_utterance", False)):
            eng.var_spans_enable(on)
            for _ in range(3):
                eng.forward_wave_var(x, lens, win, out=y)
            sync_all()
            t0 = time.perf_counter()
            for _ in range(12):
                eng.forward_wave_var(x, lens, win, out=y)
            sync_all()
            el = (time.perf_counter() - t0) / 12
            var[key] = {"ms_per_call": round(el * 1e3, 4), "frames_per_s": round(world * frames / el, 1),
                        "per_frame_cost_rel_256": round(el * 1e3 / frames / base, 3)}
    finally:
        eng.var_spans_enable(True)          # the engine is shared with the later legs: never leave the A/B switch off
    res["48x2-10s_var"] = var
    del x, y
    max_over_ranks(last, "cuda")
    torch.cuda.empty_cache()
    return {"workload": "offline wave->wave at other batch shapes, per GPU (B x clip length)", **res}


def train_prepare(rank, B=512, seconds=4.0, storage="f32"):
    """Everything of the train leg that can fail on ONE rank alone (allocations: the 16-29 GiB activation workspace,
    the model, the batch) plus one LOCAL, side-effect-free forward + backward with no collective, so that the ranks can
    agree on success before the first gradient all-reduce."""
    import torch
    from gtcrn_micro_amd.train import make_training, synthetic_mix, validate_step
    L = int(seconds * 16000)
    torch.manual_seed(43)                                     # identical initial weights on every rank
    model, opt, sched, loss_func = make_training(device="cuda")
    model.train()
    if storage != "f32":
        if not hasattr(model, "set_activation_storage"):
            return None                                       # this build has no such storage variant
        model.set_activation_storage(storage)
    noisy, clean = synthetic_mix(B, samples=L, seed=43 + rank)
    # forward + loss + backward on this rank's own shard, then everything it touched (running statistics, gradients) is
    # put back: NO optimizer / scheduler step, so the replicas still hold identical weights and Adam state when the
    # data-parallel steps start (a full local step here made every rank apply a different update first)
    validate_step(model, loss_func, noisy, clean)
    torch.cuda.synchronize()
    return {"model": model, "opt": opt, "sched": sched, "loss": loss_func, "noisy": noisy, "clean": clean,
            "B": B, "T": 1 + L // 256, "seconds": seconds, "storage": storage}


def train_run(ctx, world, sync_all, max_over_ranks, steps=5, warmup=2):
    """BASELINE configs[3]: full train steps, data parallel over utterance shards with ONE collective per step."""
    import torch
    import gtcrn_micro_amd as G
    from gtcrn_micro_amd.train import train_step
    model, opt, sched, loss_func, noisy, clean = (ctx[k] for k in ("model", "opt", "sched", "loss", "noisy", "clean"))
    B, T, seconds, storage = ctx["B"], ctx["T"], ctx["seconds"], ctx["storage"]
    if world > 1:
        # what DistributedDataParallel does at construction (train.py:88): every replica starts from rank 0's
        # parameters and buffers, whatever happened before
        from gtcrn_micro_amd.train import broadcast_parameters
        broadcast_parameters(model)
    for _ in range(warmup):
        train_step(model, opt, sched, loss_func, noisy, clean, world_size=world)
    sync_all()
    stats = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, gn = train_step(model, opt, sched, loss_func, noisy, clean, world_size=world, stats=stats)
    sync_all()
    el = max_over_ranks(time.perf_counter() - t0, "cuda") / steps
    # the exchange step alone (buffer broadcast + gradient all-reduce), from device events around it; 0 at N = 1
    xch = [a.elapsed_time(b) for a, b in stats.get("exchange_events", [])]
    allreduce_ms = round(sum(xch) / len(xch), 4) if xch else 0.0
    ws = G.Trainer.workspace_bytes(B, T, storage) if storage != "f32" else G.Trainer.workspace_bytes(B, T)
    hbm_bytes = TRAIN_HBM_BYTES_PER_STEP.get(storage)
    if hbm_bytes is not None:
        hbm_bytes = hbm_bytes * (B * T) / (512 * 251)          # the counters were taken at B = 512 x 4 s
    roof = None if hbm_bytes is None else {
        "bound": "hbm", "achieved": round(hbm_bytes / el / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(hbm_bytes / el / 1e9 / HBM_PEAK_GBS, 4), "traffic": hbm_bytes,
        "traffic_source": f"{TRAIN_HBM_SOURCE_FILE}: rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE of the profiled build on the "
                          "builder's box (a constant of that build, scaled by B x T; NOT measured by this run)",
        "note": "whole train step: counter-measured HBM bytes per step over this run's step time"}
    return {
        "workload": f"train step, B={B} clips/GPU x {seconds:g} s (T={T}), saved activations {storage}, fp32 "
                    "accumulate + master weights, Adam, clip 3.0, synthetic DNS-style mixes",
        "parallelism": f"dp{world}: utterance shards + one all-reduce of the gradient buffer per step"
                       + (" + DDP-style buffer broadcast from rank 0" if world > 1 else ""),
        "ms_per_step": round(el * 1e3, 3), "frames_per_s": round(world * B * T / el, 1),
        "steps": steps, "warmup": warmup, "dtype": storage,
        "hbm_TB_per_s": None if hbm_bytes is None else round(hbm_bytes / el / 1e12, 3),
        "roofline": roof, "allreduce_ms": allreduce_ms,
        "workspace_GB": round(ws / 2 ** 30, 2), "loss": float(loss), "grad_norm": float(gn),
        "_rate_keys": ["frames_per_s"], "_time_keys": ["ms_per_step"],
    }


def train_leg(rank, world, sync_all, max_over_ranks, B=512, seconds=4.0, steps=5, warmup=2, storage="f32"):
    """Single-process form (`--mode train`, world == 1 callers)."""
    ctx = train_prepare(rank, B, seconds, storage)
    if ctx is None:
        raise AttributeError(f"no activation storage {storage!r} in this build")
    return train_run(ctx, world, sync_all, max_over_ranks, steps, warmup)


def guarded_train_leg(storage, rank, world, get_line, B=512):
    """The train leg under run_leg's rules: phase 1 = train_prepare on every rank (local; may fail alone), ONE gather of
    the outcomes, phase 2 = the timed steps with the gradient all-reduce only if EVERY rank got through phase 1."""
    import torch
    box = {}

    def prepare(sync_local, record):
        box["ctx"] = train_prepare(rank, B=B, storage=storage)
        record(0.0)
        return {"present": box["ctx"] is not None}

    r1 = run_leg(f"train_{storage}_prepare", prepare, rank, world, get_line)
    if r1 is not None and "error" in r1:
        box.clear()
        torch.cuda.empty_cache()
        return {"error": r1["error"], "phase": "prepare (no collective had been issued)"}
    # every rank runs the same build: either all have the variant or none
    if box.get("ctx") is None:
        return None
    res = run_leg(f"train_{storage}", lambda sync_local, record: train_run(box["ctx"], world, sync_local, record),
                  rank, world, get_line)
    box.clear()
    torch.cuda.empty_cache()
    return res


def train_main(args):
    """`--mode train`: only the train-step line (not the headline metric)."""
    import torch
    import torch.distributed as dist
    from gtcrn_micro_amd.sharding import init_distributed, max_over_ranks
    rank, local_rank, world = init_distributed(None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    import __graft_entry__ as graft
    if rank == 0:
        graft.build()
    if world > 1:
        dist.barrier()

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    B = args.batch if args.batch is not None else 512         # configs[3]: 512 clips per GPU
    r = train_leg(rank, world, sync_all, max_over_ranks, B=B, seconds=args.seconds, steps=args.steps,
                  warmup=args.warmup, storage=args.train_storage)
    if rank == 0:
        print(json.dumps({
            "metric": "train frames/sec (STFT x2 -> forward -> HybridLoss -> backward -> all-reduce -> clip -> Adam)",
            "value": r["frames_per_s"], "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": r["dtype"], "data": "synthetic DNS-style mixes",
            "config": {"workload": r["workload"], "parallelism": r["parallelism"]},
            "workspace_GB": r["workspace_GB"], "hbm_TB_per_s": r["hbm_TB_per_s"],
            "loss": r["loss"], "grad_norm": r["grad_norm"]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None,
                    help="clips per GPU (default: 256 for the headline, configs[1]; 512 for --mode train, configs[3])")
    ap.add_argument("--seconds", type=float, default=4.0, help="clip length")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps each; the line reports the median one (and min / max)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the stream/train/quant legs (profiling runs of the headline path)")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer (default): the headline metric plus the secondary objects.  train: configs[3] only")
    ap.add_argument("--train-storage", choices=["f32", "bf16", "bf16_saves", "bf16_grads"], default="f32")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args, argv)                         # before torch / HIP are touched in this process

    # one process per GPU: keep this rank's host threads (launches, pinned staging, the folder driver's reader / writer)
    # on the NUMA node its GPU hangs off.  Done from sysfs BEFORE torch / HIP are imported (threads created later inherit
    # the mask); nothing is re-executed.  A single-process run is left alone (its cpu_baseline leg sweeps host threads).
    numa = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not os.environ.get("GTCRN_NO_NUMA_BIND"):
        import importlib.util
        sp = importlib.util.spec_from_file_location("gtcrn_sharding_early",
                                                    os.path.join(ROOT, "gtcrn_micro_amd", "sharding.py"))
        early = importlib.util.module_from_spec(sp)
        sp.loader.exec_module(early)
        numa = early.bind_rank_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")))

    import numpy as np
    import torch
    import torch.distributed as dist

    from gtcrn_micro_amd.sharding import init_distributed, max_over_ranks
    if args.mode == "train":
        return train_main(args)
    shim = _load_shim()
    rank, local_rank, world = init_distributed("gloo" if shim else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a mislabelled line")
    dev = "cpu" if shim else "cuda"
    if not shim:
        torch.cuda.set_device(local_rank)

    params = np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32)
    B = args.batch if args.batch is not None else 256
    L = int(args.seconds * 16000)
    T = 1 + L // 256
    frames_per_step = B * T
    torch.manual_seed(43 + rank)                               # the reference's seed (train.py:27)
    wave = (torch.randn(B, L, device=dev) * 0.1).contiguous()
    win = torch.hann_window(512).pow(0.5).to(dev)              # infer.py:65
    out = torch.empty((B, 256 * (T - 1)), device=dev)
    if shim:
        eng = shim.make_engine(params, local_rank, args)
    else:
        import __graft_entry__ as graft
        if rank == 0:
            graft.build()
        if world > 1:
            dist.barrier()
        from gtcrn_micro_amd import Engine
        eng = Engine(params, local_rank)
    eng.reserve(B, T)

    def sync_all():
        if world > 1:
            dist.barrier()
        if not shim:
            torch.cuda.synchronize()

    # pre-spin (untimed): >= 0.5 s of the same work so the clocks have settled before anything is measured
    t_spin = time.perf_counter()
    while not shim and time.perf_counter() - t_spin < 0.6:
        for _ in range(20):
            eng.forward_wave(wave, win, out=out)
        torch.cuda.synchronize()
    # W warm-up steps with HIP events around every kernel: picks the dominant kernel
    eng.timing_enable(True)
    for _ in range(max(args.warmup, 1)):
        eng.forward_wave(wave, win, out=out)
    sync_all()
    kern_warm = eng.timing_read()
    dom = max((k for k in kern_warm if k in MAC_PER_FRAME), key=lambda k: kern_warm[k][0])
    # the timed region keeps only the dominant kernel's events (on the launch stream): each event pair costs a
    # few microseconds of dispatch gap, one pair per kernel per step would be ~4 % of the step
    eng.timing_enable(True, only=dom)
    # R timed regions of EXACTLY K steps each, every one bracketed by barrier + synchronize on both sides and priced
    # with the slowest rank; the line reports min / median / max and `value` is the MEDIAN region (a single 20-50 ms
    # region is one sample of a box whose clocks move by a few percent)
    regions, host_enqueue = [], []
    for _ in range(max(args.repeats, 1)):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.forward_wave(wave, win, out=out)
        host_enqueue.append((time.perf_counter() - t0) / args.steps)   # the calls returned; nothing was waited for
        sync_all()
        mine = time.perf_counter() - t0
        regions.append((max_over_ranks(mine, dev), mine))
    dom_ms, dom_launches = eng.timing_read()[dom]
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    med = order[(len(order) - 1) // 2]                         # the (lower) median region: an actually measured one
    elapsed = regions[med][0]                                  # the slowest rank defines the step time
    per_rank_s = [x for x in gather_ranks(regions[med][1], world)]   # a straggler shows up in one line
    host_all = gather_ranks(sorted(host_enqueue)[len(host_enqueue) // 2] * 1e6, world)
    numa_all = gather_ranks(numa, world) if world > 1 else None      # per rank: {"numa_node", "cpus", "bound"} or None
    region_ms = sorted(r[0] / args.steps * 1e3 for r in regions)
    # the per-kernel split: ONE separate pass of K steps with an event pair around every kernel (all kernel_ms
    # values come from here; its step time is reported next to the headline's)
    eng.timing_enable(True)
    sync_all()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        eng.forward_wave(wave, win, out=out)
    sync_all()
    split_ms_per_step = (time.perf_counter() - t1) / args.steps * 1e3
    kern = eng.timing_read()
    eng.timing_enable(False)
    assert bool(torch.isfinite(out).all())

    line = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * frames_per_step * args.steps / elapsed
        flops_launch = 2.0 * MAC_PER_FRAME[dom] * frames_per_step
        achieved = flops_launch / (dom_ms * 1e-3) / 1e12
        traffic, traffic_source = None, None
        for tag in (ROUND_TAG, "r05", "r04", "r03", "r02", "r01"):
            tf = os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.json")
            if os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get(dom, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
                if traffic is not None:
                    traffic_source = (f"profiles/{tag}_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                      "this command on the builder's box (committed profile) -- a constant of the build, NOT "
                                      "measured by the run that printed this line")
                    break
        path_flop = (2.0 * MODEL_MAC_PER_FRAME + FFT_FLOP_PER_FRAME) * frames_per_step
        sum_ms = sum(v[0] for v in kern.values())
        roof = {
            "bound": "mfma", "kernel": dom, "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "traffic_source": traffic_source,
            "flop_per_launch": flops_launch, "avg_launch_ms": round(dom_ms, 4), "launches": dom_launches,
            # whole path against the same peak, from the headline's own step time (not from a sum of event times)
            "path_tflops": round(path_flop / (ms_per_step * 1e-3) / 1e12, 3),
            "path_frac": round(path_flop / (ms_per_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "hbm_frac_at_boundary": round(value / world * 2048 / 1e9 / HBM_PEAK_GBS, 6),
        }
        if dom in MFMA_PER_FRAME:
            # what the kernel EXECUTES on the matrix cores (frac above prices the algorithmic fp32 FLOPs against the fp32
            # matrix peak, as the metric defines the work): zero padding of the 8->16 / 16->8 pointwise and K = 15
            # tiles on the fp32 MFMA, and the dense 3x3 six times over on the bf16 pipe
            f32x = 2048.0 * MFMA_PER_FRAME[dom] * frames_per_step
            bfx = 16384.0 * BF16_MFMA_PER_FRAME.get(dom, 0.0) * frames_per_step
            roof["executed_fp32_mfma_flop_per_launch"] = f32x
            roof["executed_bf16_mfma_flop_per_launch"] = bfx
            roof["executed_tflops"] = round((f32x + bfx) / (dom_ms * 1e-3) / 1e12, 1)
            if bfx:
                roof["executed_bf16_frac_of_bf16_peak"] = round(bfx / (dom_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)
                roof["note"] = ("dense 3x3 and de_convs.3 on v_mfma_f32_16x16x32_bf16 from an exact 3-way bf16 split "
                                "of both operands (6 products, fp32 accumulate; held to the float64 graph by "
                                "tests/test_gpu_precision.py); pointwise convs on the fp32 MFMA")
        line = {
            "metric": "16 kHz frames/sec (STFT->mask->iSTFT)",
            "value": round(value, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not shim else "synthetic (TEST SHIM: control-flow test on CPU, not a measurement)",
            "config": {"workload": f"offline wave->wave, B={B} clips/GPU x {args.seconds:g} s @16 kHz "
                                   f"(T={T} frames), fp32, shipped checkpoint weights",
                       "batch_per_gpu": B, "frames_per_step_per_gpu": frames_per_step,
                       "parallelism": f"{world} independent utterance shards, no data-path collective"},
            "timed_region_s": round(elapsed, 4),
            "repeats": {"regions": len(regions), "steps_per_region": args.steps,
                        "ms_per_step": {"min": round(region_ms[0], 4), "median": round(elapsed / args.steps * 1e3, 4),
                                        "max": round(region_ms[-1], 4)},
                        "note": "value / ms_per_step / timed_region_s are the median region's; the roofline's launch "
                                "average runs over all regions"},
            "per_rank_ms_per_step": {"min": round(min(per_rank_s) / args.steps * 1e3, 4),
                                     "max": round(max(per_rank_s) / args.steps * 1e3, 4),
                                     "all": [round(x / args.steps * 1e3, 4) for x in per_rank_s]},
            "rtf_per_stream": round((elapsed / args.steps) / (B * args.seconds) * 1.0, 9),
            "numa_binding": numa_all,
            # CPU time one rank spends INSIDE forward_wave per step (argument checks, ctypes, six launches): the calls
            # are asynchronous, so this is what N processes on one host contend with, not the step time
            "host_us_per_step": round(max(host_all), 1),
            "host_us_per_step_per_rank": [round(x, 1) for x in host_all],
            "roofline": roof,
            "kernel_ms": {k: round(v[0], 4) for k, v in kern.items()},
            "sum_kernel_ms": round(sum_ms, 4),
            "kernel_ms_note": f"all from ONE separate pass of {args.steps} steps with an event pair around every "
                              f"kernel (that pass: {split_ms_per_step:.4f} ms/step); roofline.avg_launch_ms is "
                              f"{dom}'s alone inside the headline timed region",
        }
    get_line = lambda: line

    def put(name, val, into=None):                             # results join the line as they arrive: a watchdog
        if line is not None and val is not None:               # print carries every leg that finished
            (line if into is None else line.setdefault(into, {}))[name] = val

    if shim and hasattr(shim, "secondary_legs"):
        # TEST ONLY: stand-in legs that fail or hang on one rank exercise run_leg / Watchdog on CPU over gloo
        for name, (fn, timeout_s) in shim.secondary_legs(rank, world).items():
            put(name, run_leg(name, fn, rank, world, get_line, timeout_s))
    if not shim and not args.no_secondary:
        # the host-link leg FIRST: it is what a serving process does from its start, and the duplex DMA rate depends on what
        # the process did before -- after the streaming-capacity leg (40 GB of state allocated and freed) both directions
        # at once run at 28 GB/s each and the served step takes 2.0-2.6 ms; in a fresh process 48 GB/s each and 1.47 ms
        # (tools/io_overlap_probe.py; gpurun_out/r05p: the same io_leg behind each of the other legs)
        put("io", run_leg("io", lambda sync_local, record: io_leg(eng, wave, win, out, world, sync_local, record,
                                                                  steps=args.steps), rank, world, get_line))
        put("stream", run_leg("stream", lambda sync_local, record: stream_leg(eng, world, sync_local, record),
                              rank, world, get_line))
        put("stream_capacity", run_leg("stream_capacity", lambda sync_local, record: stream_capacity_leg(
            eng, world, sync_local, record), rank, world, get_line))
        put("batch_sweep", run_leg("batch_sweep", lambda sync_local, record: batch_sweep_leg(
            eng, win, world, sync_local, record), rank, world, get_line))
        eng.reserve(B, T)
        eng.forward_wave(wave, win, out=out)

        def folder(sync_local, record):
            r = folder_leg(rank, world, local_rank)
            record(r["wall_s"])
            return r
        put("folder_driver", run_leg("folder_driver", folder, rank, world, get_line))
        try:
            from gtcrn_micro_amd import quant
        except ImportError:
            quant = None
        if quant is not None:
            put("quant", run_leg("quant", lambda sync_local, record: quant.bench_leg(
                params, local_rank, wave, win, world, sync_local, record, steps=args.steps), rank, world, get_line))
        del eng, wave, out
        torch.cuda.empty_cache()
        for storage in ("f32", "bf16", "bf16_saves", "bf16_grads"):
            put(storage, guarded_train_leg(storage, rank, world, get_line), into="train")
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not shim:
            line["cpu_baseline"] = cpu_baseline(params)
            # a reported baseline, not the target: the CPU leg moves 2x from host to host on the pool, so the ratio says
            # no more than ">> 100x" (north_star's bar) -- it lives inside the object, not at the top of the line
            line["cpu_baseline"]["gpu_over_cpu"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
            hw = line["cpu_baseline"].get("host_wide") or {}
            if hw.get("value"):
                line["cpu_baseline"]["gpu_over_cpu_host_wide"] = round(line["value"] / hw["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        line = None                                            # printed: a late watchdog must not print it again
        with Watchdog(60.0, rank, lambda: None, "final barrier"):   # the line is out: nothing left to lose
            dist.barrier()
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
