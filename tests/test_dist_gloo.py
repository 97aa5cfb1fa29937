"""Multi-rank control flow on CPU (gloo, world_size 2): sharding, rendezvous on 127.0.0.1, barriers,
max-over-ranks timing and the aggregate bench line.  The data path has no collective."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions_exactly():
    from gtcrn_micro_amd.sharding import shard_range
    for n in (0, 1, 7, 256, 257, 1024):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from gtcrn_micro_amd.sharding import init_distributed, shard_range, max_over_ranks, sum_over_ranks
rank, local_rank, world = init_distributed("gloo")
assert world == 2 and dist.get_backend() == "gloo"
lo, hi = shard_range(257, world, rank)
# every rank processes its own shard of utterances; the union must be the whole list, disjoint
mine = torch.zeros(257, dtype=torch.int64); mine[lo:hi] = 1
dist.all_reduce(mine)
assert int(mine.min()) == 1 and int(mine.max()) == 1
assert max_over_ranks(1.0 + rank) == 2.0
assert sum_over_ranks(hi - lo) == 257
dist.barrier()
if rank == 0:
    print(json.dumps({"ok": True, "shard0": [lo, hi]}))
dist.destroy_process_group()
"""


def _torchrun(args, timeout=300, extra_env=None, nproc=2):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_shard_and_reduce(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(_WORKER)
    r = _torchrun([str(w), ROOT])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"ok": True, "shard0": [0, 129]}


def _check_bench_line(stdout, steps, warmup, world=2):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # only rank 0 prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == "weak"
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    T = 1 + 8000 // 256
    assert abs(d["value"] - world * 4 * T * steps / (d["ms_per_step"] * steps / 1e3)) / d["value"] < 1e-3
    assert "cpu_baseline" not in d and "TEST SHIM" in d["data"]
    assert "stream" not in d and "train" not in d       # the secondary legs need the HIP engine
    pr = d["per_rank_ms_per_step"]                      # a straggler is visible in the line
    assert len(pr["all"]) == world and pr["min"] <= pr["max"] and abs(pr["max"] - d["ms_per_step"]) < 1e-3
    # host-side cost of a step per rank (what N processes on one host contend with) and where each rank was bound
    assert len(d["host_us_per_step_per_rank"]) == world and d["host_us_per_step"] == max(d["host_us_per_step_per_rank"])
    assert len(d["numa_binding"]) == world and all(set(b) == {"numa_node", "cpus", "bound", "physical_gpu"} for b in d["numa_binding"])
    return d


_SHIM_ENV = {"GTCRN_BENCH_TEST_SHIM": os.path.join(ROOT, "tests", "bench_shim.py")}
_BENCH_ARGS = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--seconds", "0.5"]


def test_bench_control_flow_two_ranks():
    """bench.py's N>1 path under an external launcher (the driver's form): barrier + max over ranks + one JSON
    line from rank 0."""
    r = _torchrun([os.path.join(ROOT, "bench.py")] + _BENCH_ARGS, extra_env=_SHIM_ENV)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_bench_line(r.stdout, 3, 1)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun starts two ranks itself and reports n_gpus == 2."""
    env = dict(os.environ, OMP_NUM_THREADS="1", **_SHIM_ENV)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + _BENCH_ARGS, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_bench_line(r.stdout, 3, 1)


def test_bench_refuses_mislabelled_world():
    """WORLD_SIZE=2 from the launcher but --gpus 4: non-zero exit, no JSON line."""
    args = list(_BENCH_ARGS)
    args[1] = "4"
    r = _torchrun([os.path.join(ROOT, "bench.py")] + args, extra_env=_SHIM_ENV)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_has_no_builtin_fake_engine():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "--cpu-stub" not in src and "class _Stub" not in src


_GRAD_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from gtcrn_micro_amd.sharding import init_distributed
from gtcrn_micro_amd.train import allreduce_gradients
from gtcrn_micro_amd.models.gtcrn_micro import GTCRNMicro
rank, local_rank, world = init_distributed("gloo")
m = GTCRNMicro()                       # parameter containers only: nothing is computed on the CPU
g = torch.Generator().manual_seed(100 + rank)
# (1) gradients that did not come from the HIP backward (hand-set): packed, reduced, unpacked
for p in m.parameters():
    if p.requires_grad:
        p.grad = torch.randn(p.shape, generator=g)
mine = torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).clone()
n = allreduce_gradients(m, world)
both = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
want = (both[0] + both[1]) / 2
got = torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])
ok = bool(torch.allclose(got, want, atol=1e-6)) and n == 19014
# (2) the layout the HIP backward leaves: every .grad is a view of ONE blob in the canonical order ->
#     the blob itself is all-reduced in place, no cat / copy-back
m._flatten(torch.device("cpu"))
blob = torch.randn(44938, generator=g)
m._grad_flat = blob
for p, (off, numel, shape) in zip(m._train_params, m._train_slices):
    p.grad = blob[off:off + numel].view(shape)
mine = blob.clone()
ptr = blob.data_ptr()
n2 = allreduce_gradients(m, world)
both = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
ok2 = bool(torch.allclose(blob, (both[0] + both[1]) / 2, atol=1e-6)) and n2 == 44938 and blob.data_ptr() == ptr
ok2 = ok2 and all(p.grad.data_ptr() == ptr + 4 * off for p, (off, _, _) in zip(m._train_params, m._train_slices))
dist.barrier()
if rank == 0:
    print(json.dumps({"ok": ok, "floats": n, "ok_flat": ok2, "floats_flat": n2}))
dist.destroy_process_group()
"""


def test_gradient_allreduce_two_ranks(tmp_path):
    """The one exchange step of data-parallel training (train.py:87-88): the gradients are averaged over the ranks
    as one contiguous message -- the kernel's own gradient blob when the .grads are views of it."""
    w = tmp_path / "grad_worker.py"
    w.write_text(_GRAD_WORKER)
    r = _torchrun([str(w), ROOT])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"ok": True, "floats": 19014, "ok_flat": True, "floats_flat": 44938}


_BUF_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from gtcrn_micro_amd.sharding import init_distributed
from gtcrn_micro_amd.train import broadcast_buffers, allreduce_gradients
from gtcrn_micro_amd.models.gtcrn_micro import GTCRNMicro
rank, local_rank, world = init_distributed("gloo")
torch.manual_seed(7)                   # identical initial weights on every rank, like the trainer
m = GTCRNMicro()
m._flatten(torch.device("cpu"))        # the layout a train-mode forward leaves: everything is a view of one blob
g = torch.Generator().manual_seed(100 + rank)
params_before = torch.cat([p.detach().reshape(-1) for p in m.parameters() if p.requires_grad]).clone()
# what a local-batch-statistics forward does on each rank: DIFFERENT running statistics per rank
with torch.no_grad():
    for name, b in m.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            b.add_(torch.randn(b.shape, generator=g))
        elif name.endswith("num_batches_tracked"):
            b.add_(1 + rank)
n = broadcast_buffers(m)
stats = torch.cat([b.reshape(-1).double() for name, b in m.named_buffers()
                   if name.endswith("running_mean") or name.endswith("running_var") or name.endswith("num_batches_tracked")])
both = [torch.zeros_like(stats) for _ in range(world)]
dist.all_gather(both, stats)
same = bool(torch.equal(both[0], both[1]))
params_after = torch.cat([p.detach().reshape(-1) for p in m.parameters() if p.requires_grad])
untouched = bool(torch.equal(params_before, params_after))
nbt = int(next(b for name, b in m.named_buffers() if name.endswith("num_batches_tracked")))
# a model whose tensors are NOT views of a flat blob takes the buffer-by-buffer path
m2 = GTCRNMicro()
with torch.no_grad():
    for name, b in m2.named_buffers():
        if name.endswith("running_var"):
            b.fill_(2.0 + rank)
broadcast_buffers(m2)
rv = float(next(b for name, b in m2.named_buffers() if name.endswith("running_var"))[0])
dist.barrier()
if rank == 0:
    print(json.dumps({"floats": n, "same": same, "params_untouched": untouched, "nbt": nbt, "rv_unflat": rv}))
dist.destroy_process_group()
"""


def test_running_statistics_follow_rank0_two_ranks(tmp_path):
    """DDP's broadcast_buffers (the reference's default, train.py:88) on the explicit all-reduce path: after
    broadcast_buffers() both ranks hold rank 0's running statistics and counters bit for bit, the trainable
    tensors are untouched, and the 1 348 running-stat floats travelled as one message (+ the 46 counters)."""
    w = tmp_path / "buf_worker.py"
    w.write_text(_BUF_WORKER)
    r = _torchrun([str(w), ROOT])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"floats": 1348, "same": True, "params_untouched": True, "nbt": 1, "rv_unflat": 2.0}


_DP_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from gtcrn_micro_amd.sharding import init_distributed
from gtcrn_micro_amd.train import allreduce_gradients, broadcast_buffers, broadcast_parameters
from gtcrn_micro_amd.models.gtcrn_micro import GTCRNMicro
rank, local_rank, world = init_distributed("gloo")
torch.manual_seed(500 + rank)          # DIFFERENT initial weights per rank: whatever happened before the dp steps
m = GTCRNMicro()
m._flatten(torch.device("cpu"))
def everything():
    return torch.cat([t.detach().reshape(-1).double() for t in list(m.parameters()) + list(m.buffers())])
def same_on_all_ranks(v):
    both = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(both, v)
    return bool(torch.equal(both[0], both[1]))
differ_before = not same_on_all_ranks(everything())
n = broadcast_parameters(m)            # DDP's construction-time broadcast (train.py:88)
equal_after_broadcast = same_on_all_ranks(everything())
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(900 + rank)
equal_after_steps = True
for step in range(3):                  # the dp step of train_step() with the HIP backward replaced by rank-specific
    with torch.no_grad():              # gradients (and rank-specific running statistics from the local batches)
        for name, b in m.named_buffers():
            if name.endswith("running_mean") or name.endswith("running_var"):
                b.add_(torch.randn(b.shape, generator=g) * 0.01)
    broadcast_buffers(m)
    blob = torch.randn(44938, generator=g)
    m._grad_flat = blob
    for p, (off, numel, shape) in zip(m._train_params, m._train_slices):
        p.grad = blob[off:off + numel].view(shape)
    allreduce_gradients(m, world)
    torch.nn.utils.clip_grad_norm_(m.parameters(), 3.0)
    opt.step()
    equal_after_steps = equal_after_steps and same_on_all_ranks(everything())
adam = torch.cat([st["exp_avg"].reshape(-1).double() for st in opt.state.values()])
adam_equal = same_on_all_ranks(adam)
# a model whose tensors are NOT views of a flat blob takes the tensor-by-tensor path
torch.manual_seed(700 + rank)
m2 = GTCRNMicro()
broadcast_parameters(m2)
v2 = torch.cat([t.detach().reshape(-1).double() for t in list(m2.parameters()) + list(m2.buffers())])
unflat_equal = same_on_all_ranks(v2)
dist.barrier()
if rank == 0:
    print(json.dumps({"differ_before": differ_before, "floats": n, "equal_after_broadcast": equal_after_broadcast,
                      "equal_after_steps": equal_after_steps, "adam_equal": adam_equal, "unflat_equal": unflat_equal}))
dist.destroy_process_group()
"""


def test_replicas_stay_bit_equal_through_data_parallel_steps(tmp_path):
    """The dp semantics the train leg claims: replicas that start from DIFFERENT weights are made identical by the
    construction-time broadcast (DDP, train.py:88), and stay identical -- parameters, running statistics, Adam
    moments, bit for bit -- through steps of buffer broadcast -> rank-specific gradients -> ONE all-reduce -> clip ->
    Adam.  (bench.py's train leg used to run a full LOCAL step per rank before the dp steps: the replicas diverged.)"""
    w = tmp_path / "dp_worker.py"
    w.write_text(_DP_WORKER)
    r = _torchrun([str(w), ROOT])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"differ_before": True, "floats": 44938, "equal_after_broadcast": True,
                                "equal_after_steps": True, "adam_equal": True, "unflat_equal": True}


def test_train_leg_prepare_has_no_optimizer_step():
    """bench.py's train_prepare validates the allocations with validate_step (forward + backward, state restored), never
    with a full train_step: a local optimizer step per rank is what made the replicas diverge."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "train_prepare")
    called = {c.func.id for c in ast.walk(fn) if isinstance(c, ast.Call) and isinstance(c.func, ast.Name)}
    assert "validate_step" in called and "train_step" not in called


def test_secondary_leg_failing_on_one_rank_does_not_hang_or_lose_the_line():
    """Rank 1 raises inside a secondary leg before any collective: the leg reports the error per rank, the legs before
    and after it complete (priced with the slower rank's time) and the headline is intact."""
    r = _torchrun([os.path.join(ROOT, "bench.py")] + _BENCH_ARGS, extra_env=dict(_SHIM_ENV, GTCRN_BENCH_TEST_LEGS="fail1"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = _check_bench_line(r.stdout, 3, 1)
    assert d["second"] == {"error": {"1": "RuntimeError('simulated out-of-memory on rank 1')"}}
    for name in ("first", "third"):
        assert d[name]["units_per_s"] == 1000.0 * 0.5 and d[name]["ms"] == 20.0      # rank 0 timed 10 ms, rank 1 20 ms
        assert d[name]["per_rank_timed_s"] == {"min": 0.01, "max": 0.02}


def test_secondary_leg_hanging_on_one_rank_is_cut_by_the_watchdog():
    """Rank 1 never returns from a secondary leg (the shape of a rank stuck in, or dead before, a collective): after the
    leg's time limit rank 0 prints the line as it stands -- headline and the finished leg -- and every rank exits with a
    non-zero status (a watchdog on a process that may hold the GPU must not report success; no re-exec)."""
    r = _torchrun([os.path.join(ROOT, "bench.py")] + _BENCH_ARGS, timeout=120,
                  extra_env=dict(_SHIM_ENV, GTCRN_BENCH_TEST_LEGS="hang1"))
    assert r.returncode != 0            # a rank hung: the line is kept, but the launcher / driver see a FAILED run
    d = _check_bench_line(r.stdout, 3, 1)
    assert "timed out after 6 s" in d["secondary_errors"]["second"]
    assert d["first"]["ms"] == 20.0 and "third" not in d


# ---- world_size 8: what an 8-GPU node runs (no such node is available to the builder: CPU ranks over gloo) ---------

_WORKER8 = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from gtcrn_micro_amd.sharding import init_distributed, shard_range, max_over_ranks, sum_over_ranks
rank, local_rank, world = init_distributed("gloo")
assert world == 8 and dist.get_backend() == "gloo"
ok = True
for n in (1027, 256 * 8, 5, 0):        # BASELINE configs[1] x 8, a ragged list, fewer items than ranks, nothing
    lo, hi = shard_range(n, world, rank)
    mine = torch.zeros(max(n, 1), dtype=torch.int64); mine[lo:hi] = 1
    dist.all_reduce(mine)
    ok = ok and (n == 0 or (int(mine[:n].min()) == 1 and int(mine[:n].max()) == 1))
    ok = ok and sum_over_ranks(hi - lo) == n
assert max_over_ranks(1.0 + rank) == 8.0
dist.barrier()
if rank == 0:
    print(json.dumps({"ok": bool(ok)}))
dist.destroy_process_group()
"""


def test_eight_ranks_shard_and_reduce(tmp_path):
    w = tmp_path / "worker8.py"
    w.write_text(_WORKER8)
    r = _torchrun([str(w), ROOT], nproc=8)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]) == {"ok": True}


def test_bench_control_flow_eight_ranks():
    """`bench.py --gpus 8` as the driver launches it (external torchrun, one rank per GPU): one line, 8 entries in
    per_rank_ms_per_step / host_us_per_step_per_rank / numa_binding."""
    args = list(_BENCH_ARGS)
    args[1] = "8"
    r = _torchrun([os.path.join(ROOT, "bench.py")] + args, extra_env=_SHIM_ENV, nproc=8, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_bench_line(r.stdout, 3, 1, world=8)


_SCP_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from gtcrn_micro_amd.sharding import init_distributed, shard_range
from gtcrn_micro_amd import infer
rank, local_rank, world = init_distributed("gloo")
enh = sys.argv[2]
names = sorted(f"mix_fileid_{k}.wav" for k in range(int(sys.argv[3])))
def fake_shard(noisy_dir, clean_dir, enh_dir, checkpoint, device, max_batch, rank, world, barrier, **kw):
    # the host side of _enhance_shard without the GPU: this rank's slice of the SORTED list, written as its scp files
    lo, hi = shard_range(len(names), world, rank)
    os.makedirs(enh_dir, exist_ok=True)
    inf = [(n[:-4], os.path.join(enh_dir, n[:-4] + "_enh.wav")) for n in names[lo:hi]]
    ref = [(n[:-4], os.path.join(clean_dir, "clean_" + n.split("_", 1)[1])) for n in names[lo:hi]]
    for fname, lst in (("inf.scp", inf), ("ref.scp", ref)):
        with open(os.path.join(enh_dir, f"{fname}.rank{rank}"), "w") as f:
            for uid, p in lst:
                f.write(f"{uid} {p}\n")
    return inf, ref
infer._enhance_shard = fake_shard
def agree(ok):
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())
infer.enhance_folder("noisy", "clean", enh, "ck", rank=rank, world=world, barrier=dist.barrier, agree=agree)
dist.barrier()
if rank == 0:
    got = [l.split()[0] for l in open(os.path.join(enh, "inf.scp"))]
    ref = [l.split()[0] for l in open(os.path.join(enh, "ref.scp"))]
    print(json.dumps({"merged_sorted": got == [n[:-4] for n in names], "ref_same_order": ref == got, "n": len(got)}))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("nfiles", [37, 5])
def test_folder_driver_scp_merge_eight_ranks(tmp_path, nfiles):
    """infer.py:113-119 writes ONE inf.scp / ref.scp in sorted file order; with eight ranks each writes its shard's
    lists and rank 0 merges them in rank order = sorted order, also when some ranks hold no file at all."""
    w = tmp_path / "scp_worker.py"
    w.write_text(_SCP_WORKER)
    r = _torchrun([str(w), ROOT, str(tmp_path / "enh"), str(nfiles)], nproc=8)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"merged_sorted": True, "ref_same_order": True, "n": nfiles}


_DP8_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from gtcrn_micro_amd.sharding import init_distributed
from gtcrn_micro_amd.train import allreduce_gradients, broadcast_buffers, broadcast_parameters, _loss_slot
from gtcrn_micro_amd.models.gtcrn_micro import GTCRNMicro
from gtcrn_micro_amd import _lib
rank, local_rank, world = init_distributed("gloo")
torch.manual_seed(500 + rank)          # DIFFERENT initial weights per rank
m = GTCRNMicro()
m._flatten(torch.device("cpu"))
def everything():
    return torch.cat([t.detach().reshape(-1).double() for t in list(m.parameters()) + list(m.buffers())])
def same_on_all_ranks(v):
    allv = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(allv, v)
    return all(bool(torch.equal(allv[0], x)) for x in allv[1:])
broadcast_parameters(m)
equal_after_broadcast = same_on_all_ranks(everything())
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(900 + rank)
bufmask = torch.ones(44938, dtype=torch.bool)
for p, (off, numel, shape) in zip(m._train_params, m._train_slices):
    bufmask[off:off + numel] = False
equal_after_steps, losses_ok, slot_clean = True, True, True
for step in range(2):
    broadcast_buffers(m)
    blob = torch.randn(44938, generator=g)
    blob[bufmask] = 0.0                # what the HIP backward leaves: buffers have no gradient
    m._grad_flat = blob
    for p, (off, numel, shape) in zip(m._train_params, m._train_slices):
        p.grad = blob[off:off + numel].view(shape)
    loss = torch.tensor(10.0 * step + rank)            # this rank's loss of the step
    mean = allreduce_gradients(m, world, loss=loss)    # rides in a zero slot of the SAME message
    losses_ok = losses_ok and abs(float(mean) - (10.0 * step + (world - 1) / 2)) < 1e-5
    slot_clean = slot_clean and float(blob[_loss_slot(m, blob.device)]) == 0.0 and bool((blob[bufmask] == 0).all())
    torch.nn.utils.clip_grad_norm_(m.parameters(), 3.0)
    opt.step()
    equal_after_steps = equal_after_steps and same_on_all_ranks(everything())
# gradients that are NOT views of the blob: the loss is appended to the packed message
for p in m.parameters():
    if p.requires_grad:
        p.grad = torch.randn(p.shape, generator=g)
m._grad_flat = None
mean2 = allreduce_gradients(m, world, loss=torch.tensor(float(rank)))
packed_ok = abs(float(mean2) - (world - 1) / 2) < 1e-5
dist.barrier()
if rank == 0:
    print(json.dumps({"equal_after_broadcast": equal_after_broadcast, "equal_after_steps": equal_after_steps,
                      "losses_ok": losses_ok, "slot_clean": slot_clean, "packed_ok": packed_ok}))
dist.destroy_process_group()
"""


def test_eight_replicas_stay_bit_equal_and_the_loss_rides_in_the_gradient_message(tmp_path):
    """World size 8: construction-time broadcast, then dp steps (buffer broadcast -> rank-specific gradients -> ONE
    all-reduce -> clip -> Adam) keep all eight replicas bit-equal; the step's loss travels in a zero slot of the
    gradient blob and comes back as the all-rank mean (reduce_value, train.py:268-269) with NO collective of its own;
    the slot is zero again afterwards (it is a buffer's gradient position)."""
    w = tmp_path / "dp8_worker.py"
    w.write_text(_DP8_WORKER)
    r = _torchrun([str(w), ROOT], nproc=8, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"equal_after_broadcast": True, "equal_after_steps": True, "losses_ok": True,
                                "slot_clean": True, "packed_ok": True}


def test_train_step_issues_no_separate_loss_collective():
    """train_step's only collectives are broadcast_buffers and allreduce_gradients (which carries the loss)."""
    import ast
    src = open(os.path.join(ROOT, "gtcrn_micro_amd", "train.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "train_step")
    attrs = {c.func.attr for c in ast.walk(fn) if isinstance(c, ast.Call) and isinstance(c.func, ast.Attribute)}
    assert "all_reduce" not in attrs and "reduce" not in attrs
    names = {c.func.id for c in ast.walk(fn) if isinstance(c, ast.Call) and isinstance(c.func, ast.Name)}
    assert {"allreduce_gradients", "broadcast_buffers"} <= names
