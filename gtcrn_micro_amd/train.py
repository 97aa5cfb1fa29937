"""Counterpart of the reference's train step (train.py:239-288) on MI355X.

The outer loop stays PyTorch-ROCm, exactly as north_star asks: Hann-window STFTs of the noisy/clean
batch on the device (HIP kernel), ``model(noisy_spec)`` in train mode (HIP forward, saved
activations), HybridLoss (torch ops), ``loss.backward()`` (HIP backward through the autograd node of
models/gtcrn_micro.py), ``clip_grad_norm_(3.0)``, Adam, warm-up-cosine schedule.  Data parallel
training shards independent utterances over the ranks; the ONE exchange step is the all-reduce of the
19 014 gradient floats (train.py:87-88 wraps the model in DistributedDataParallel; here either do the
same, or call ``allreduce_gradients`` which moves them as one contiguous buffer over RCCL)."""
import math

import torch

from . import _lib
from .loss import HybridLoss
from .models.gtcrn_micro import GTCRNMicro
from .utils.scheduler import LinearWarmupCosineAnnealingLR


def synthetic_mix(batch, samples=64000, seed=43, device="cuda"):
    """DNS-style synthetic mixes: speech-like harmonic ``clean`` (5 harmonics of a
    100-300 Hz f0, 4 Hz amplitude modulation, peak 0.5) + low-passed Gaussian noise at an SNR drawn
    from U(-5, 20) dB; returns (noisy, clean), both (batch, samples) fp32 in [-1, 1]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = torch.arange(samples, dtype=torch.float32) / 16000.0
    f0 = 100.0 + 200.0 * torch.rand(batch, 1, generator=g)
    ph = 2 * math.pi * torch.rand(batch, 5, generator=g)
    clean = sum(torch.sin(2 * math.pi * (h + 1) * f0 * t + ph[:, h:h + 1]) / (h + 1) for h in range(5))
    clean = clean * (0.6 + 0.4 * torch.sin(2 * math.pi * 4.0 * t))
    clean = 0.5 * clean / clean.abs().amax(dim=1, keepdim=True)
    noise = torch.randn(batch, samples, generator=g)
    noise = torch.nn.functional.avg_pool1d(noise[:, None], 5, 1, 2)[:, 0]          # crude low-pass
    snr = -5.0 + 25.0 * torch.rand(batch, 1, generator=g)
    scale = clean.pow(2).mean(1, keepdim=True).sqrt() / (noise.pow(2).mean(1, keepdim=True).sqrt() * 10 ** (snr / 20))
    noisy = (clean + scale * noise).clamp(-1, 1)
    return noisy.to(device), clean.to(device)


def make_training(config=None, device="cuda", fused_optimizer=True):
    """Model, Adam, scheduler and loss with the reference's defaults (conf/cfg_train_DNS3.yaml).
    fused_optimizer: ``utils.optim.FlatAdam`` (clip + Adam as one HIP launch over the flat blobs; same update rule and
    state_dict as torch.optim.Adam) instead of ``torch.optim.Adam`` itself (train.py:90)."""
    cfg = {"lr": 1e-3, "warmup_steps": 25000, "decay_until_step": 250000, "max_lr": 1e-3, "min_lr": 1e-6}
    cfg.update(config or {})
    model = GTCRNMicro().to(device)
    if fused_optimizer:
        from .utils.optim import FlatAdam
        opt = FlatAdam(model, lr=cfg["lr"])
    else:
        opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"])
    sched = LinearWarmupCosineAnnealingLR(opt, cfg["warmup_steps"], cfg["decay_until_step"], cfg["max_lr"],
                                          cfg["min_lr"])
    return model, opt, sched, HybridLoss().to(device)


def _flat_gradient_blob(model):
    """The contiguous gradient blob of the last HIP backward if every ``p.grad`` is (still) a view of it, else None."""
    flat = getattr(model, "_grad_flat", None)
    slices = getattr(model, "_train_slices", None)
    params = getattr(model, "_train_params", None)
    if flat is None or slices is None or params is None:
        return None
    base = flat.data_ptr()
    for p, (off, numel, _) in zip(params, slices):
        if p.grad is None or p.grad.data_ptr() != base + 4 * off or not p.grad.is_contiguous():
            return None
    return flat


def _buffer_index(model, device):
    """Positions (int64, on ``device``) of the BatchNorm running statistics inside the canonical blob: 92 tensors,
    1 348 floats, interleaved with the trainable tensors in state_dict order.  Built once per model and device."""
    cache = getattr(model, "_buffer_index_cache", None)
    if cache is None or cache.device != torch.device(device):
        idx = [torch.arange(off, off + numel) for name, numel, off in _lib.param_table()
               if name.endswith("running_mean") or name.endswith("running_var")]
        cache = torch.cat(idx).to(device)
        model._buffer_index_cache = cache
    return cache


def broadcast_buffers(model, src=0):
    """DistributedDataParallel's ``broadcast_buffers=True`` (the default the reference trains with, train.py:88) on the
    explicit all-reduce path: every rank's BatchNorm running statistics and ``num_batches_tracked`` counters follow
    rank ``src``.  BatchNorm uses LOCAL batch statistics in the forward (plain BN, not SyncBN, models/gtcrn_micro.py:159),
    so without this the 1 348 running-stat floats (and 46 counters) drift apart per rank and a checkpoint or an eval on rank != 0 differs
    from what the reference's DDP run would hold.  When the parameters live in the flat blob the statistics are
    gathered into ONE message (5.4 KB), broadcast and scattered back; otherwise buffer by buffer.  Returns the floats
    on the wire."""
    import torch.distributed as dist
    m = model.module if hasattr(model, "module") else model
    flat = getattr(m, "_flat", None)
    if flat is not None and m._flat_ok(flat.device):
        idx = _buffer_index(m, flat.device)
        msg = flat.index_select(0, idx)
        dist.broadcast(msg, src=src)
        flat.index_copy_(0, idx, msg)
        dist.broadcast(m._nbt_flat, src=src)
        for k, (eng, _) in list(m._engines.items()):
            m._engines[k] = (eng, None)    # the eval engines re-fold on next use
        return int(msg.numel())
    n = 0
    for b in m.buffers():
        dist.broadcast(b.data, src=src)
        n += b.numel() if b.dtype.is_floating_point else 0
    return n


def broadcast_parameters(model, src=0):
    """What DistributedDataParallel does when it wraps a module (train.py:88): every rank's parameters AND buffers are
    overwritten with rank ``src``'s, so the replicas start identical whatever each process did before.  ONE message
    when the model lives in the flat blob (44 938 floats + the 46 counters), tensor by tensor otherwise.  Returns the
    floats on the wire."""
    import torch.distributed as dist
    m = model.module if hasattr(model, "module") else model
    flat = getattr(m, "_flat", None)
    if flat is not None and m._flat_ok(flat.device):
        dist.broadcast(flat, src=src)
        dist.broadcast(m._nbt_flat, src=src)
        for k, (eng, _) in list(m._engines.items()):
            m._engines[k] = (eng, None)    # the eval engines re-fold on next use
        return int(flat.numel())
    n = 0
    with torch.no_grad():
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src)
            n += t.numel() if t.dtype.is_floating_point else 0
    return n


def _loss_slot(model, device):
    """A float of the gradient blob that is ALWAYS zero after a backward (the first BatchNorm running-mean position:
    buffers receive no gradient): the train loss rides there through the gradient all-reduce."""
    slot = getattr(model, "_loss_slot_cache", None)
    if slot is None:                      # from the host-side table: no device read, no synchronisation
        slot = next(off for name, _, off in _lib.param_table() if name.endswith("running_mean"))
        model._loss_slot_cache = slot
    return slot


def allreduce_gradients(model, world_size, loss=None):
    """The one exchange step of data-parallel training (the reference gets it from DDP, train.py:87-88): average
    the gradients over the ranks as ONE contiguous message.  After a HIP backward the 248 ``.grad``s are views of
    the kernel's own gradient blob (canonical layout: 19 014 trainable floats + the zero slots of the buffers,
    44 938 floats = 180 KB; latency-bound over xGMI), so the blob is all-reduced in place with no packing; gradients
    that came from elsewhere (hand-set, accumulated) are packed and unpacked.

    ``loss`` (optional 0-d tensor): the reference averages the step's loss over the ranks with a collective of its own
    (``reduce_value(loss)``, train.py:268-269, utils/distributed_utils.py:69-79).  Here it travels in one of the
    25 924 zero slots of the same message -- no second collective: the slot is written before the all-reduce, read and
    cleared after it.  Returns the all-rank mean loss (a 0-d tensor) when ``loss`` is given, else the floats on the
    wire.

    Why the whole 180 KB blob and not only its 19 014 trainable floats (76 KB): the trainable tensors are interleaved
    with the BatchNorm buffers in the canonical (state_dict) order, so the short message needs a gather kernel before
    and a scatter kernel after the collective (two launches, ~10 us) to save ~100 KB on a ring whose per-step cost is
    its latency (tens of microseconds over xGMI), not its bytes (100 KB at ~50 GB/s effective = 2 us)."""
    import torch.distributed as dist
    m = model.module if hasattr(model, "module") else model
    flat = _flat_gradient_blob(m)
    if flat is not None:
        slot = _loss_slot(m, flat.device) if loss is not None else None
        if slot is not None:
            flat[slot:slot + 1].copy_(loss.detach().reshape(1))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / world_size)
        if slot is None:
            return flat.numel()
        mean_loss = flat[slot].clone()
        flat[slot:slot + 1].zero_()          # the slot is a buffer's gradient position again: zero
        return mean_loss
    ps = [p for p in m.parameters() if p.grad is not None]
    parts = [p.grad.reshape(-1) for p in ps]
    if loss is not None:
        parts.append(loss.detach().reshape(1).to(parts[0].dtype))
    flat = torch.cat(parts)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world_size
    off = 0
    for p in ps:
        p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
        off += p.numel()
    return flat[off].clone() if loss is not None else flat.numel()


def validate_step(model, loss_func, noisy, clean, window=None):
    """Forward + HybridLoss + backward of one batch with NO lasting effect: no optimizer or scheduler step, the
    BatchNorm running statistics and counters are put back, the gradients are cleared.  For callers that must find out
    whether a train step FITS (workspace allocation, kernels) before committing to a collective step: replicas that
    each ran a full local step first would apply different updates and start data-parallel training from different
    weights.  Returns the loss."""
    m = model.module if hasattr(model, "module") else model
    win = window if window is not None else torch.hann_window(512, device=noisy.device)
    # every parameter and buffer of the mirror is a view of ONE device blob once it has run (GTCRNMicro._flatten): the
    # statistics are saved and put back as two copies, not as 2 x 139 (which showed up as ~280 copyBuffer launches per
    # prepared train leg in the kernel statistics)
    flat = None
    if noisy.is_cuda and hasattr(m, "_flatten"):
        if not m._flat_ok(noisy.device):
            m._flatten(noisy.device)
        flat = (m._flat.detach().clone(), m._nbt_flat.detach().clone())
    else:
        saved = [b.detach().clone() for b in m.buffers()]
    loss = loss_func(model(_lib.stft(noisy, win)), _lib.stft(clean, win))
    loss.backward()
    with torch.no_grad():
        if flat is not None and m._flat_ok(noisy.device):
            m._flat.copy_(flat[0])          # (the weights in it are untouched: no optimizer step ran)
            m._nbt_flat.copy_(flat[1])
        elif flat is not None:
            raise _lib.GtcrnError("validate_step: the model's tensors stopped being views of its flat blob during the step")
        else:
            for b, v in zip(m.buffers(), saved):
                b.copy_(v)
    for p in m.parameters():
        p.grad = None
    return loss.detach()


def validate_batch(model, loss_func, noisy, clean, world_size=1, window=None):
    """One iteration of Trainer._validation_epoch (train.py:301-342): Hann-window STFTs of the pair, the eval-mode
    forward under ``torch.inference_mode`` (the fused inference kernels, BatchNorm running statistics), HybridLoss
    (value only: no gradient is formed), the all-rank mean of the loss when world_size > 1 (``reduce_value``,
    train.py:330-331), and the enhanced waveform by the Hann-window iSTFT, padded / cropped to the clean length as
    train.py:343-350 does before scoring.  Returns (loss, enhanced_wave).  PESQ scoring (third-party C, CPU) is the
    caller's, as in the reference.  The model must be in ``.eval()`` mode (Trainer.train() switches it, train.py:414-417)."""
    m = model.module if hasattr(model, "module") else model
    if m.training:
        raise _lib.GtcrnError("validate_batch: call model.eval() first (the validation epoch runs the eval-mode model, "
                              "train.py:414-417)")
    win = window if window is not None else torch.hann_window(512, device=noisy.device)
    with torch.inference_mode():
        noisy_spec = _lib.stft(noisy, win, frame_major=True)
        clean_spec = _lib.stft(clean, win, frame_major=True)
        enhanced = model(noisy_spec)
        loss = loss_func(enhanced, clean_spec)
        if world_size > 1:
            import torch.distributed as dist
            loss = loss.clone()
            dist.all_reduce(loss, op=dist.ReduceOp.SUM)
            loss = loss / world_size
        wave = _lib.istft(enhanced, win)
        L = clean.shape[-1]
        if wave.shape[-1] < L:
            wave = torch.nn.functional.pad(wave, (0, L - wave.shape[-1]))
        elif wave.shape[-1] > L:
            wave = wave[..., :L]
    return loss, wave


def train_step(model, optimizer, scheduler, loss_func, noisy, clean, clip_grad_norm_value=3.0, world_size=1,
               window=None, stats=None):
    """One iteration of Trainer._train_epoch (train.py:244-288); returns (loss, grad_norm) as 0-d device tensors (no host
    synchronisation inside the step); with world_size > 1 the loss is the mean over the ranks (train.py:268-269).
    stats (optional dict): device events around the exchange steps (buffer broadcast, gradient all-reduce) are
    appended to stats["exchange_events"] as (start, end) pairs, so a caller can report what the collectives cost."""
    dev = noisy.device
    win = window if window is not None else torch.hann_window(512, device=dev)      # train.py:252 (Hann, not sqrt)
    timed = stats is not None and world_size > 1 and dev.type == "cuda"

    def exchange(fn):
        if not timed:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        stats.setdefault("exchange_events", []).append((a, b))

    if world_size > 1 and model.training:
        # DDP semantics (broadcast_buffers=True): the buffers follow rank 0 at the start of every forward, so the
        # running statistics a rank holds are rank 0's of the previous step plus its own update of this step
        exchange(lambda: broadcast_buffers(model))
    # (B,257,T,2)-shaped like torch.stft's, in the library's frame-major memory order: the model output and the loss gradient
    # follow the input's order, so every kernel of the step walks whole 2 KB frames
    noisy_spec = _lib.stft(noisy, win, frame_major=True)
    clean_spec = _lib.stft(clean, win, frame_major=True)
    enhanced = model(noisy_spec)
    loss = loss_func(enhanced, clean_spec)
    optimizer.zero_grad()
    loss.backward()
    if world_size > 1:
        # the gradient all-reduce also carries the loss: what comes back is the all-rank mean, as the reference's
        # reduce_value(loss) returns it (train.py:268-269) -- one collective, not two
        box = {}
        exchange(lambda: box.__setitem__("loss", allreduce_gradients(model, world_size, loss=loss)))
        loss = box["loss"]
    if hasattr(optimizer, "step_clipped"):
        gn = optimizer.step_clipped(clip_grad_norm_value)      # clip_grad_norm_ + Adam: one launch (utils/optim.py)
    else:
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad_norm_value)
        optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss.detach(), gn


def save_checkpoint(path, model, optimizer, scheduler, epoch):
    """The reference's checkpoint dict (train.py:200-216): {"epoch", "optimizer", "scheduler", "model"}, with the
    model's 388-key state_dict on the CPU (DDP's wrapper is peeled like ``self.model.module`` there), so a
    checkpoint written here loads into the reference and the other way round."""
    m = model.module if hasattr(model, "module") else model
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    torch.save({"epoch": int(epoch), "optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict(),
                "model": sd}, path)


def load_checkpoint(path, model, optimizer=None, scheduler=None, map_location="cpu"):
    """Counterpart of Trainer._resume_checkpoint (train.py:223-237) and of infer.py:39-41; tolerates DDP's
    ``module.`` prefix (stream_onnx.py:45-47).  Returns the next epoch."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    m = model.module if hasattr(model, "module") else model
    m.load_state_dict(ck["model"])
    if optimizer is not None and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    if scheduler is not None and "scheduler" in ck:
        scheduler.load_state_dict(ck["scheduler"])
    return int(ck.get("epoch", 0)) + 1
