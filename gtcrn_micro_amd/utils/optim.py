"""``clip_grad_norm_`` + ``torch.optim.Adam`` of the reference's train step (train.py:90, 282-285) as ONE HIP launch.

After a HIP backward every ``p.grad`` of a ``GTCRNMicro`` is a view of one gradient blob and every parameter a view of
one parameter blob (canonical layout of the C ABI).  PyTorch's own clip + Adam still walk the 248 views: a norm kernel
per tensor, a stack, a foreach multiply, then the foreach Adam -- about 300 launches and 200 small buffer copies per step
(``profiles/r04_train_f32_kernel_stats.csv``: 202 ``copyBuffer`` + 101 elementwise launches, ~0.9 ms of a 30 ms step) for
19 014 floats.  ``FlatAdam`` keeps both Adam moments in two more flat blobs and calls ``gtcrn_clip_adam_step`` once.

It IS a ``torch.optim.Optimizer``: ``param_groups`` (so the reference's scheduler drives ``lr``), ``zero_grad``,
``state`` with per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` entries (views of the flat moments), and a
``state_dict`` that ``torch.optim.Adam.load_state_dict`` accepts and vice versa -- the reference's checkpoints
(train.py:200-237) carry the optimizer state.  The update rule is torch.optim.Adam's (no amsgrad, L2 weight decay)."""
import torch

from .. import _lib


def _grad_blob(model):
    """The gradient blob of the last HIP backward if every ``p.grad`` is still a view of it, else None."""
    flat = getattr(model, "_grad_flat", None)
    if flat is None or model._train_slices is None:
        return None
    base = flat.data_ptr()
    for p, (off, _, _) in zip(model._train_params, model._train_slices):
        if p.grad is None or p.grad.data_ptr() != base + 4 * off or not p.grad.is_contiguous():
            return None
    return flat


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        m = model.module if hasattr(model, "module") else model
        if not hasattr(m, "_flatten"):
            raise _lib.GtcrnError("FlatAdam drives a gtcrn_micro_amd GTCRNMicro (its parameters live in one flat blob)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("FlatAdam: invalid hyper-parameter")
        # the group carries every key torch.optim.Adam's groups carry, so the two optimizers can load each other's state
        defaults = dict(torch.optim.Adam([torch.zeros(1)]).defaults)
        defaults.update(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(list(m.parameters()), defaults)
        self._model = m
        self._m = self._v = self._mask = None
        self._uniform = True      # every state entry carries the same step count (the normal case: no per-step check)
        self._wsbuf = None        # the launch pair's own workspace (ticket + partial sums): two optimizers stepping on
                                  # different streams of one device must not share one
        self._nstep = 0
        self._steps = []          # one 0-d CPU tensor per parameter, like torch.optim.Adam's state["step"] (a tensor shared
                                  # by all entries would be incremented 248 times per step by an Adam that loaded this state)

    # -- flat moments -------------------------------------------------------------------------------------------------
    def _adopt(self, device):
        """(Re)builds the flat moment blobs on `device`, carrying over whatever the per-parameter state holds (a loaded
        state_dict), and points the state entries at views of them."""
        m = self._model
        n = _lib.NPARAM_FLOATS
        fm = torch.zeros(n, device=device, dtype=torch.float32)
        fv = torch.zeros(n, device=device, dtype=torch.float32)
        mask = torch.zeros(n, device=device, dtype=torch.float32)
        step = None
        own = []                  # the step count each entry brings along (None: no entry)
        for p, (off, numel, shape) in zip(m._train_params, m._train_slices):
            st = self.state[p]
            if "exp_avg" in st:
                fm[off:off + numel].copy_(st["exp_avg"].reshape(-1))
                fv[off:off + numel].copy_(st["exp_avg_sq"].reshape(-1))
                own.append(int(float(st["step"])))
                step = own[-1] if step is None else step
            else:
                own.append(None)
            mask[off:off + numel] = 1.0
        # (a loaded state without entries starts from step 0, like a fresh torch.optim.Adam: keeping an older count would
        # bias-correct zeroed moments with the wrong power)
        self._nstep = step if step is not None else 0
        self._steps = []
        # (entries of one torch.optim.Adam run carry one count; a state with several -- a parameter was frozen for a while --
        # keeps them, and step_clipped then checks that the parameters it steps agree)
        self._uniform = all(o is None or o == self._nstep for o in own)
        for (p, (off, numel, shape)), o in zip(zip(m._train_params, m._train_slices), own):
            st = self.state[p]
            st["step"] = torch.tensor(float(self._nstep if o is None else o))
            st["exp_avg"] = fm[off:off + numel].view(shape)
            st["exp_avg_sq"] = fv[off:off + numel].view(shape)
            self._steps.append(st["step"])
        self._m, self._v, self._mask = fm, fv, mask

    def _ws(self, device):
        if self._wsbuf is None or self._wsbuf.device != device:
            self._wsbuf = _lib.clip_adam_workspace(device)
        return self._wsbuf

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._m = None                              # the loaded per-parameter tensors are adopted at the next step

    # -- the step -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step_clipped(self, max_norm=0.0):
        """clip_grad_norm_(parameters, max_norm) followed by Adam.step(), one launch; returns the total gradient norm
        (0-d device tensor) like clip_grad_norm_.  max_norm <= 0: no clipping (the norm is still returned)."""
        if len(self.param_groups) != 1:
            raise _lib.GtcrnError("FlatAdam: one parameter group (the reference trains with one, train.py:90)")
        m = self._model
        flat = m._flat
        if flat is None or not flat.is_cuda or not m._flat_ok(flat.device):
            raise _lib.GtcrnError("FlatAdam.step: the model's tensors are not views of its flat device blob (run a train-"
                                  "mode forward / backward first; there is no CPU path)")
        if self._m is None or self._m.device != flat.device:
            self._adopt(flat.device)
        g = _grad_blob(m)
        packed = g is None
        mask = self._mask
        live = self._steps
        if packed:                                  # gradients from elsewhere (hand-set, accumulated): pack, run, unpack
            g = torch.zeros_like(flat)
            frozen = [i for i, p in enumerate(m._train_params) if p.grad is None]
            if frozen:
                # torch.optim.Adam and clip_grad_norm_ skip a parameter without a gradient: neither its weight (weight
                # decay included), nor its moments, nor its step count move.  The kernel updates where the mask is set.
                mask = mask.clone()
                for i in frozen:
                    off, numel, _ = m._train_slices[i]
                    mask[off:off + numel] = 0.0
                fz = set(frozen)
                live = [st for i, st in enumerate(self._steps) if i not in fz]
                self._uniform = False
            for p, (off, numel, _) in zip(m._train_params, m._train_slices):
                if p.grad is not None:
                    g[off:off + numel].copy_(p.grad.reshape(-1))
        grp = self.param_groups[0]
        if not self._uniform and live and any(float(st) != float(live[0]) for st in live):
            raise _lib.GtcrnError("FlatAdam: the parameters being stepped have different step counts (one was frozen for a "
                                  "while): the one-launch step bias-corrects with ONE count -- use torch.optim.Adam")
        if not live:
            return torch.zeros((), device=flat.device)
        self._nstep = int(float(live[0])) + 1
        torch._foreach_add_(live, 1)                # (host tensors: what a state_dict carries)
        norm = torch.empty(2, device=flat.device, dtype=torch.float32)
        _lib.clip_adam_step(flat, g, self._m, self._v, mask, self._nstep, grp["lr"], grp["betas"], grp["eps"],
                            grp["weight_decay"], max_norm, norm, ws=self._ws(flat.device))
        if packed and max_norm > 0:
            for p, (off, numel, _) in zip(m._train_params, m._train_slices):
                if p.grad is not None:
                    p.grad.copy_(g[off:off + numel].view_as(p.grad))
        m._opt_serial += 1                          # the kernel wrote the weights in place: inference engines re-fold
        self._opt_called = True                     # (what LRScheduler's wrapper of step() records)
        return norm[0]

    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.step_clipped(0.0)
        return loss
