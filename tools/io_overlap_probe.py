#!/usr/bin/env python3
"""Which forms of a copy-in || kernels || copy-out pipeline actually overlap on this box?  bench.py's `io` object
reports the served rate; this probe is where its pipeline shape was chosen.  One JSON object per run.
    python tools/io_overlap_probe.py
    HSA_ENABLE_SDMA=0 python tools/io_overlap_probe.py     # copies as blit kernels instead of SDMA engines"""
import json
import os
import queue
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# BIND=local / remote / none: run on the CPUs of the GPU's NUMA node, of another node, or wherever the scheduler puts us
# (decided BEFORE torch / HIP are imported: the pinned staging buffers are then allocated from that node)
import importlib.util
_sp = importlib.util.spec_from_file_location("gtcrn_sharding_early", os.path.join(ROOT, "gtcrn_micro_amd", "sharding.py"))
_sh = importlib.util.module_from_spec(_sp)
_sp.loader.exec_module(_sh)
BIND = os.environ.get("BIND", "none")
_numa = {"mode": BIND, "gpu_numa_nodes": _sh.gpu_numa_nodes()}
if BIND == "local":
    _numa.update(_sh.bind_rank_to_gpu_numa(0))
elif BIND == "remote" and _numa["gpu_numa_nodes"]:
    other = [n for n in range(8) if n != _numa["gpu_numa_nodes"][0] and
             os.path.exists(f"/sys/devices/system/node/node{n}/cpulist")]
    if other:
        cpus = sorted(set(_sh.parse_cpulist(open(f"/sys/devices/system/node/node{other[0]}/cpulist").read())) &
                      set(os.sched_getaffinity(0)))
        if cpus:
            os.sched_setaffinity(0, cpus)
            _numa.update({"numa_node": other[0], "cpus": len(cpus), "bound": True})
import numpy as np
import torch
from gtcrn_micro_amd import Engine

params = np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32)
B, L = 256, 64000
T = 1 + L // 256
engs = [Engine(params, 0) for _ in range(2)]
for e in engs:
    e.reserve(B, T)
win = torch.hann_window(512).pow(0.5).cuda()
din = [torch.randn(B, L, device="cuda") * 0.1 for _ in range(2)]
dout = [torch.empty(B, 256 * (T - 1), device="cuda") for _ in range(2)]
hin = [torch.empty(din[0].shape, pin_memory=True) for _ in range(2)]
hout = [torch.empty(dout[0].shape, pin_memory=True) for _ in range(2)]
for h in hin:
    h.copy_(din[0])
s_in, s_cmp, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
STEPS = 24


def timed(fn, n=STEPS):
    fn(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 4)


def independent(n):          # no dependencies at all: the ceiling
    for i in range(n):
        with torch.cuda.stream(s_in):
            din[1].copy_(hin[0], non_blocking=True)
        with torch.cuda.stream(s_cmp):
            engs[0].forward_wave(din[0], win, out=dout[0])
        with torch.cuda.stream(s_out):
            hout[0].copy_(dout[1], non_blocking=True)


def three_streams_events(n):  # bench r05a form: one issuing thread, cross-stream events in front of copies and kernels
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_cmp = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(s_in):
            s_in.wait_event(ev_cmp[k])
            din[k].copy_(hin[k], non_blocking=True)
            ev_in[k].record()
        with torch.cuda.stream(s_cmp):
            s_cmp.wait_event(ev_in[k])
            s_cmp.wait_event(ev_out[k])
            engs[0].forward_wave(din[k], win, out=dout[k])
            ev_cmp[k].record()
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev_cmp[k])
            hout[k].copy_(dout[k], non_blocking=True)
            ev_out[k].record()


def two_lanes_two_engines(n):  # lane = stream doing in -> kernels -> out in order; even / odd batches; no events
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(lanes[k]):
            din[k].copy_(hin[k], non_blocking=True)
            engs[k].forward_wave(din[k], win, out=dout[k])
            hout[k].copy_(dout[k], non_blocking=True)


def two_lanes_one_engine(n):   # same, ONE engine: its kernels are ordered across the lanes by kernel-to-kernel events
    ev = [torch.cuda.Event() for _ in range(2)]
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(lanes[k]):
            din[k].copy_(hin[k], non_blocking=True)
            lanes[k].wait_event(ev[k ^ 1])
            engs[0].forward_wave(din[k], win, out=dout[k])
            ev[k].record()
            hout[k].copy_(dout[k], non_blocking=True)


def three_threads(n):          # a thread per stage, dependencies resolved by host-side event waits (no cross-stream waits)
    free = threading.Semaphore(2)
    q1, q2 = queue.Queue(), queue.Queue()

    def t_in():
        for i in range(n):
            free.acquire()
            k = i & 1
            with torch.cuda.stream(s_in):
                din[k].copy_(hin[k], non_blocking=True)
                e = torch.cuda.Event()
                e.record()
            e.synchronize()
            q1.put(k)
        q1.put(None)

    def t_cmp():
        while True:
            k = q1.get()
            if k is None:
                q2.put(None)
                return
            with torch.cuda.stream(s_cmp):
                engs[0].forward_wave(din[k], win, out=dout[k])
                e = torch.cuda.Event()
                e.record()
            e.synchronize()
            q2.put(k)

    def t_out():
        while True:
            k = q2.get()
            if k is None:
                return
            with torch.cuda.stream(s_out):
                hout[k].copy_(dout[k], non_blocking=True)
                e = torch.cuda.Event()
                e.record()
            e.synchronize()
            free.release()
    ths = [threading.Thread(target=f) for f in (t_in, t_cmp, t_out)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()


for _ in range(30):
    engs[0].forward_wave(din[0], win, out=dout[0])
torch.cuda.synchronize()
def pair(n):                  # the two copies alone, concurrently
    for i in range(n):
        with torch.cuda.stream(s_in):
            din[1].copy_(hin[0], non_blocking=True)
        with torch.cuda.stream(s_out):
            hout[0].copy_(dout[1], non_blocking=True)


def one_in(n):
    for i in range(n):
        with torch.cuda.stream(s_in):
            din[1].copy_(hin[0], non_blocking=True)


res = {"HSA_ENABLE_SDMA": os.environ.get("HSA_ENABLE_SDMA"), "steps": STEPS, "numa": _numa,
       "in_only_ms": timed(one_in), "in_and_out_ms": timed(pair), "in_and_out_again_ms": timed(pair)}
for name, fn in (("independent_ceiling", independent), ("three_streams_events", three_streams_events),
                 ("two_lanes_two_engines", two_lanes_two_engines), ("two_lanes_one_engine", two_lanes_one_engine),
                 ("three_threads_host_waits", three_threads), ("independent_ceiling_again", independent),
                 ("three_streams_events_again", three_streams_events)):
    try:
        res[name + "_ms_per_step"] = timed(fn)
    except Exception as e:
        res[name + "_ms_per_step"] = repr(e)
print(json.dumps(res))
