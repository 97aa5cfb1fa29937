#!/usr/bin/env python3
"""Streaming measurement (BASELINE config 3): N concurrent streams, one 16 ms frame per call.

    python tools/stream_bench.py [--streams 1024] [--frames 251] [--chunk 1]

Reports per-call latency (mean / p50 / p99), frame-steps/s and the real-time factor per stream
(call time / (chunk * 16 ms)).  State stays on the device in the library's ring layout
(StreamGTCRNMicro.step / gtcrn_stream_step); the spectrogram frames are resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=251)
    ap.add_argument("--chunk", type=int, default=1, help="frames per call")
    a = ap.parse_args()
    import numpy as np
    import torch
    import __graft_entry__ as graft
    graft.build()
    from gtcrn_micro_amd import Engine
    params = np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32)
    eng = Engine(params, 0)
    N, T, C = a.streams, a.frames, a.chunk
    torch.manual_seed(43)
    spec = torch.randn(N, T, 257, 2, device="cuda") * 0.3          # frame-major storage
    spec = spec.permute(0, 2, 1, 3)                                 # viewed as (N,257,T,2)
    out = torch.empty((N, T, 257, 2), device="cuda").permute(0, 2, 1, 3)
    state = eng.new_state(N)
    eng.reserve(N, C)
    for t in range(0, min(T, 8 * C), C):                            # warm-up
        eng.stream_step(state, spec[:, :, t:t + C], out=out[:, :, t:t + C])
    state = eng.new_state(N)
    torch.cuda.synchronize()
    lat = []
    t_all = time.perf_counter()
    for t in range(0, T - C + 1, C):
        t0 = time.perf_counter()
        eng.stream_step(state, spec[:, :, t:t + C], out=out[:, :, t:t + C])
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    total = time.perf_counter() - t_all
    lat = np.array(lat) * 1e3
    # the same frames offline, for a consistency check
    full = eng.forward_spec(spec[:, :, :len(lat) * C].contiguous())
    err = float((full - out[:, :, :len(lat) * C]).abs().max() / full.abs().max())
    print(json.dumps({
        "streams": N, "frames_per_call": C, "calls": len(lat),
        "latency_ms": {"mean": round(float(lat.mean()), 4), "p50": round(float(np.percentile(lat, 50)), 4),
                       "p99": round(float(np.percentile(lat, 99)), 4), "max": round(float(lat.max()), 4)},
        "frame_steps_per_s": round(N * C * len(lat) / total, 1),
        "rtf_per_stream": round(float(lat.mean()) / (C * 16.0), 5),
        "stream_vs_offline_rel_err": err,
        "state_bytes_per_stream": eng.state_bytes(),
    }))


if __name__ == "__main__":
    main()
