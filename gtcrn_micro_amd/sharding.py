"""Utterance/stream sharding over the GPUs of one node.

Inference is embarrassingly parallel over utterances (eval-mode BatchNorm uses running statistics,
so there is no cross-utterance dependence): every rank takes a contiguous, balanced shard and owns a
full copy of the 76 KB of weights.  No data-path collective exists; torch.distributed (RCCL on the
GPUs, gloo in the CPU tests) is only used for rendezvous, barriers and reducing timings/counts.
"""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_items, world, rank):
    """Contiguous balanced shard [lo, hi) of n_items for `rank` of `world` (first n % world ranks get one more)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_distributed(backend=None):
    """One process per GPU; returns (rank, local_rank, world).  127.0.0.1 rendezvous by default."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def max_over_ranks(value, device="cpu"):
    """max of a python float over all ranks (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


# ---- rank -> NUMA node of its GPU ------------------------------------------------------------------------------
# An 8-GPU MI355X node has two sockets with four GPUs behind each; a rank whose host thread (launches, pinned staging
# buffers, the folder driver's reader / writer threads) runs on the other socket pays a cross-socket hop per launch and
# per staged byte.  torchrun does not bind; the reference's mp.spawn (train.py:461-471) does not either.  This binds the
# calling process to the CPUs of its GPU's NUMA node from sysfs alone -- it must run BEFORE torch / HIP are imported
# (threads created earlier keep their old mask) and never re-executes anything.

def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0,1,2,3,8,10,11] (the kernel's cpulist format)."""
    cpus = []
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return cpus


def gpu_numa_nodes(sysfs_root="/"):
    """NUMA node of every GPU in HIP enumeration order: the KFD topology lists the nodes in the order the runtime
    enumerates them (CPU nodes have no `drm_render_minor`), and the render node's PCI device carries `numa_node`.
    [] when the topology is not readable (no GPU, container without /sys/class/kfd)."""
    base = os.path.join(sysfs_root, "sys/class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return []
    out = []
    for n in nodes:
        props = _read(os.path.join(base, str(n), "properties")) or ""
        kv = dict(line.split(None, 1) for line in props.splitlines() if len(line.split(None, 1)) == 2)
        minor = int(kv.get("drm_render_minor", "0") or 0)
        if int(kv.get("simd_count", "0") or 0) == 0 or minor <= 0:
            continue                                           # a CPU node
        numa = _read(os.path.join(sysfs_root, f"sys/class/drm/renderD{minor}/device/numa_node"))
        out.append(int(numa) if numa not in (None, "") else -1)
    return out


def _visible_index(local_rank):
    """The physical GPU index behind `local_rank`, or None when it cannot be told.  The visibility lists COMPOSE:
    HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES is its alias on ROCm) indexes INTO the devices ROCR_VISIBLE_DEVICES leaves
    visible, so local_rank is mapped through the HIP / CUDA list first and the result through the ROCR list.  A list in
    another form (UUIDs "GPU-...") or an index past a list's end gives None: the caller then leaves the CPU mask alone
    instead of pinning the rank to a guessed -- possibly the other -- socket."""
    idx = local_rank
    for names in (("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"), ("ROCR_VISIBLE_DEVICES",)):
        v = next((os.environ[n] for n in names if os.environ.get(n)), None)
        if not v:
            continue
        items = [x.strip() for x in v.split(",") if x.strip()]
        if not all(x.isdigit() for x in items) or idx >= len(items):
            return None
        idx = int(items[idx])
    return idx


def bind_rank_to_gpu_numa(local_rank, sysfs_root="/", apply=True):
    """Restricts the calling process to the CPUs of the NUMA node its GPU hangs off (intersected with the CPUs it is
    allowed to use now).  Returns a dict describing what was done -- {"numa_node", "cpus", "bound"} -- so the bench line
    can say so; a node without topology files, a GPU that reports no node (-1: single-socket box) or an empty
    intersection leave the mask alone ("bound": False).  Call it first thing in a rank's process."""
    info = {"numa_node": None, "cpus": None, "bound": False, "physical_gpu": None}
    numas = gpu_numa_nodes(sysfs_root)
    idx = _visible_index(local_rank)
    info["physical_gpu"] = idx if idx is not None else "unknown"
    if idx is None or not numas or idx >= len(numas) or numas[idx] < 0:
        return info
    node = numas[idx]
    info["numa_node"] = node
    cpus = parse_cpulist(_read(os.path.join(sysfs_root, f"sys/devices/system/node/node{node}/cpulist")))
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return info
    want = sorted(set(cpus) & set(allowed))
    if not want:
        return info
    info["cpus"] = len(want)
    if apply:
        try:
            os.sched_setaffinity(0, want)
        except OSError:
            return info
    info["bound"] = True
    return info
