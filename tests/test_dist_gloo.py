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


def _torchrun(args, timeout=300):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_shard_and_reduce(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(_WORKER)
    r = _torchrun([str(w), ROOT])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"ok": True, "shard0": [0, 129]}


def test_bench_control_flow_two_ranks():
    """bench.py's N>1 path end to end (barrier + max over ranks + one JSON line from rank 0)."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                   "--batch", "4", "--seconds", "0.5", "--cpu-stub"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # only rank 0 prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    T = 1 + 8000 // 256
    assert abs(d["value"] - 2 * 4 * T * 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 1e-3
    assert "cpu_baseline" not in d and "STUB" in d["data"]


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
dist.barrier()
if rank == 0:
    print(json.dumps({"ok": ok, "floats": n}))
dist.destroy_process_group()
"""


def test_gradient_allreduce_two_ranks(tmp_path):
    """The one exchange step of data-parallel training (train.py:87-88): the 19 014 gradient floats are
    averaged over the ranks as one contiguous buffer."""
    w = tmp_path / "grad_worker.py"
    w.write_text(_GRAD_WORKER)
    r = _torchrun([str(w), ROOT])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"ok": True, "floats": 19014}
