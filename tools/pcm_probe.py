"""Probe of the served pipeline with 16-bit PCM at the boundary (bench.py io.served_pcm16_*): the conversion kernels and the
int16 copies alone, then the three-stream pipeline in float32 and PCM16 form with the host's issue time per step."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import gtcrn_micro_amd as G
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
eng = G.Engine(np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32), 0)
B, L = 256, 64000
wave = torch.randn(B, L, device="cuda") * 0.1
win = torch.hann_window(512).pow(0.5).cuda()
dev = wave.device
hin = [torch.empty(B, L).pin_memory() for _ in range(2)]
hout = [torch.empty(B, L).pin_memory() for _ in range(2)]
din = [torch.empty(B, L, device=dev) for _ in range(2)]
dout = [torch.empty(B, L, device=dev) for _ in range(2)]
hin16 = [torch.empty(B, L, dtype=torch.int16).pin_memory() for _ in range(2)]
hout16 = [torch.empty(B, L, dtype=torch.int16).pin_memory() for _ in range(2)]
din16 = [torch.empty(B, L, dtype=torch.int16, device=dev) for _ in range(2)]
dout16 = [torch.empty(B, L, dtype=torch.int16, device=dev) for _ in range(2)]
s_in, s_cmp, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
ev_in = [torch.cuda.Event() for _ in range(2)]; ev_cmp = [torch.cuda.Event() for _ in range(2)]; ev_out = [torch.cuda.Event() for _ in range(2)]
def served(n, pcm, conv=True):
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(s_in):
            s_in.wait_event(ev_cmp[k])
            (din16 if pcm else din)[k].copy_((hin16 if pcm else hin)[k], non_blocking=True)
            ev_in[k].record()
        with torch.cuda.stream(s_cmp):
            s_cmp.wait_event(ev_in[k]); s_cmp.wait_event(ev_out[k])
            if pcm and conv: G.pcm16_to_f32(din16[k], out=din[k])
            eng.forward_wave(din[k], win, out=dout[k])
            if pcm and conv: G.f32_to_pcm16(dout[k], out=dout16[k])
            ev_cmp[k].record()
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev_cmp[k])
            (hout16 if pcm else hout)[k].copy_((dout16 if pcm else dout)[k], non_blocking=True)
            ev_out[k].record()
for name, pcm, conv in (("float32", False, True), ("pcm16", True, True), ("pcm16 copies only", True, False), ("float32", False, True), ("pcm16", True, True)):
    served(40, pcm, conv); torch.cuda.synchronize()
    t0 = time.perf_counter(); served(40, pcm, conv); th = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"{name:20s} {tt / 40 * 1e3:.4f} ms per step, host issue {th / 40 * 1e3:.4f} ms per step")
