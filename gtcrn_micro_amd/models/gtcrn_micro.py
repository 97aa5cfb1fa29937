"""Host-side mirror of the reference's ``gtcrn_micro.models.gtcrn_micro`` for MI355X.

``GTCRNMicro`` keeps the reference's public surface -- constructor arguments
(accepted and ignored, models/gtcrn_micro.py:486-504), ``forward(spec)`` on a
``(B,257,T,2)`` spectrogram (:506-532), the 388 ``state_dict`` keys of the shipped
checkpoint, ``.eval()/.to()/.parameters()`` -- but owns no arithmetic: the torch
sub-modules below are parameter containers only, and ``forward`` hands raw device
pointers to the HIP library through the C ABI (include/gtcrn_micro_hip.h).
There is no CPU path: a CPU tensor, a missing library or a non-gfx950 device raise.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import _lib


def erb_filter_bank(n_low=65, n_bands=64, nfft=512, high_lim=8000.0, fs=16000):
    """The fixed ERB bank of the reference (models/gtcrn_micro.py:35-61): ``n_bands`` triangular
    filters over bins ``n_low..nfft/2`` whose edges are equally spaced on the ERB scale
    ``21.4*log10(0.00437 f + 1)`` and rounded to FFT bins.  Returns (n_bands, nfft/2+1-n_low) fp32."""
    to_erb = lambda hz: 21.4 * np.log10(0.00437 * hz + 1)
    to_hz = lambda e: (10 ** (e / 21.4) - 1) / 0.00437
    edges = np.linspace(to_erb(n_low / nfft * fs), to_erb(high_lim), n_bands)
    b = np.round(to_hz(edges) / fs * nfft).astype(np.int32)
    bank = np.zeros((n_bands, nfft // 2 + 1), np.float32)
    eps = 1e-12

    def fall(lo, hi):   # 1 -> 0 over bins [lo, hi)
        return (hi - np.arange(lo, hi) + eps) / (hi - lo + eps)

    def rise(lo, hi):   # 0 -> 1 over bins [lo, hi)
        return (np.arange(lo, hi) - lo + eps) / (hi - lo + eps)

    bank[0, b[0]:b[1]] = fall(b[0], b[1])
    for j in range(1, n_bands - 1):
        bank[j, b[j - 1]:b[j]] = rise(b[j - 1], b[j])
        bank[j, b[j]:b[j + 1]] = fall(b[j], b[j + 1])
    bank[-1, b[-2]:b[-1] + 1] = 1 - bank[-2, b[-2]:b[-1] + 1]
    return np.abs(bank[:, n_low:])


# Every (re-)registration of a parameter, buffer or sub-module anywhere in the process bumps this epoch (global torch
# hooks, installed once): `mod.weight = nn.Parameter(...)`, parametrize / weight_norm / prune, `register_buffer`,
# swapping a sub-module.  GTCRNMicro caches the list of tensor OBJECTS behind its state_dict; an object that is
# replaced does not change any version counter, so the cache is rebuilt whenever the epoch moved.
_REG_EPOCH = [0]


def _bump_epoch(*args):
    _REG_EPOCH[0] += 1
    return None


try:
    from torch.nn.modules import module as _tmod
    _tmod.register_module_parameter_registration_hook(_bump_epoch)
    _tmod.register_module_buffer_registration_hook(_bump_epoch)
    _tmod.register_module_module_registration_hook(_bump_epoch)
    _HAVE_REG_HOOKS = True
except Exception:                     # an older torch: fall back to walking the state_dict on every call
    _HAVE_REG_HOOKS = False


class _Holder(nn.Module):
    """A node of the parameter tree; it is never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the arithmetic runs in the HIP library")


class ERB(_Holder):
    def __init__(self, erb_subband_1=65, erb_subband_2=64, nfft=512, high_lim=8000, fs=16000):
        super().__init__()
        bank = torch.from_numpy(erb_filter_bank(erb_subband_1, erb_subband_2, nfft, high_lim, fs))
        nfreqs = nfft // 2 + 1
        self.erb_subband_1 = erb_subband_1
        self.erb_fc = nn.Linear(nfreqs - erb_subband_1, erb_subband_2, bias=False)
        self.ierb_fc = nn.Linear(erb_subband_2, nfreqs - erb_subband_1, bias=False)
        self.erb_fc.weight = nn.Parameter(bank, requires_grad=False)
        self.ierb_fc.weight = nn.Parameter(bank.T.contiguous(), requires_grad=False)


class SFE_Lite(_Holder):
    def __init__(self, in_channels=3):
        super().__init__()
        self.depth_conv = nn.Conv2d(in_channels, in_channels, (1, 3), padding=(0, 1), groups=in_channels, bias=False)


class TRALite(_Holder):
    def __init__(self, channels, kernel=3):
        super().__init__()
        self.channels, self.kernel, self.L = channels, kernel, kernel - 1
        self.depth_conv = nn.Conv1d(channels, channels, kernel, groups=channels, bias=True)
        self.point_conv = nn.Conv1d(channels, channels, 1, bias=True)


class ConvBlock(_Holder):
    def __init__(self, cin, cout, kernel_size, stride, padding, groups=1, use_deconv=False, is_last=False):
        super().__init__()
        conv = nn.ConvTranspose2d if use_deconv else nn.Conv2d
        self.conv = conv(cin, cout, kernel_size, stride, padding, groups=groups)
        self.bn = nn.BatchNorm2d(cout)
        self.act = nn.Tanh() if is_last else nn.PReLU()


class GTConvBlock(_Holder):
    def __init__(self, in_channels, hidden, kernel_size, stride, padding, dilation, use_deconv=False):
        super().__init__()
        self.use_deconv = use_deconv
        conv = nn.ConvTranspose2d if use_deconv else nn.Conv2d
        self.point_conv1 = conv(in_channels // 2, hidden, 1)
        self.point_bn1 = nn.BatchNorm2d(hidden)
        self.point_act = nn.PReLU()
        # encoder: depthwise (groups = 16); decoder: dense transposed (groups = 1)  (reference :191-212)
        self.depth_conv = conv(hidden, hidden, kernel_size, stride=stride, padding=padding, dilation=dilation,
                               groups=1 if use_deconv else 16)
        self.depth_bn = nn.BatchNorm2d(hidden)
        self.depth_act = nn.PReLU()
        self.point_conv2 = conv(hidden, in_channels // 2, 1)
        self.point_bn2 = nn.BatchNorm2d(in_channels // 2)
        self.tra = TRALite(in_channels // 2)


class TCN(_Holder):
    def __init__(self, channels, kernel_size=3, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, 1)
        self.bn1 = nn.BatchNorm2d(channels)
        self.act1 = nn.PReLU()
        self.conv2 = nn.Conv2d(channels, channels, (kernel_size, 1), dilation=(dilation, 1), groups=channels)
        self.bn2 = nn.BatchNorm2d(channels)
        self.act2 = nn.PReLU()
        self.conv3 = nn.Conv2d(channels, channels, 1)
        self.bn3 = nn.BatchNorm2d(channels)
        self.act3 = nn.PReLU()


class GTCN(_Holder):
    def __init__(self, channels, n_layers=4, kernel_size=3, dilation=2):
        super().__init__()
        self.blocks = nn.ModuleList(TCN(channels, kernel_size, dilation ** i) for i in range(n_layers))


def _gt(deconv):
    return GTConvBlock(16, 16, (3, 3), stride=(1, 1), padding=(0, 1), dilation=(1, 1), use_deconv=deconv)


class Encoder(_Holder):
    def __init__(self):
        super().__init__()
        self.en_convs = nn.ModuleList([
            ConvBlock(3, 16, (1, 5), stride=(1, 2), padding=(0, 2)),
            ConvBlock(16, 16, (1, 5), stride=(1, 2), padding=(0, 2)),
            _gt(False), _gt(False), _gt(False)])


class Decoder(_Holder):
    def __init__(self):
        super().__init__()
        self.de_convs = nn.ModuleList([
            _gt(True), _gt(True), _gt(True),
            ConvBlock(16, 16, (1, 5), stride=(1, 2), padding=(0, 2), use_deconv=True),
            ConvBlock(16, 2, (1, 5), stride=(1, 2), padding=(0, 2), use_deconv=True, is_last=True)])


class Mask(_Holder):
    pass


def state_dict_to_blob(sd):
    """Reference state_dict (optionally with DDP's ``module.`` prefix, stream_onnx.py:45-47) -> the
    canonical flat fp32 blob of the C ABI (order from gtcrn_param_name)."""
    if any(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    parts = []
    for name, numel, _ in _lib.param_table():
        if name not in sd:
            raise KeyError(f"state_dict lacks '{name}'")
        t = sd[name].detach().to(device="cpu", dtype=torch.float32).reshape(-1)
        if t.numel() != numel:
            raise ValueError(f"'{name}' has {t.numel()} elements, expected {numel}")
        parts.append(t)
    return torch.cat(parts).numpy()


def load_blob_into(model, blob):
    """The inverse of ``state_dict_to_blob``: the canonical flat fp32 blob (numpy, gtcrn_param_name order) -> the
    module's parameters and running statistics (``num_batches_tracked`` counters are not part of the blob)."""
    blob = np.ascontiguousarray(blob, np.float32).ravel()
    if blob.size != _lib.NPARAM_FLOATS:
        raise ValueError(f"parameter blob must hold {_lib.NPARAM_FLOATS} floats, got {blob.size}")
    sd = model.state_dict()
    new = {name: torch.from_numpy(blob[off:off + numel].copy()).view(sd[name].shape)
           for name, numel, off in _lib.param_table()}
    res = model.load_state_dict(new, strict=False)
    # strict in everything but the counters the blob does not carry: a key or shape-table mismatch must not leave
    # tensors silently unloaded
    missing = [k for k in res.missing_keys if not k.endswith("num_batches_tracked")]
    if missing or res.unexpected_keys:
        raise KeyError(f"load_blob_into: the blob's tensor table does not match the module (missing {missing[:5]}, "
                       f"unexpected {list(res.unexpected_keys)[:5]})")
    return model


def _is_trainable(name):
    return not (name.endswith("running_mean") or name.endswith("running_var") or name.startswith("erb."))


class _TrainStep(torch.autograd.Function):
    """``model(spec)`` in train mode: the HIP train-mode forward (gtcrn_train_forward) and, for
    ``loss.backward()``, the HIP backward (gtcrn_train_backward).  The trainable tensors are inputs of
    this node, so autograd, ``clip_grad_norm_``, Adam and DistributedDataParallel see ordinary ``.grad``s
    (train.py:280-286); the spectrogram is data and receives no gradient."""

    @staticmethod
    def forward(ctx, spec, model, *params):
        tr = model._trainer(spec.device)
        out = tr.forward(model._flat, spec)
        model._nbt_flat += 1              # num_batches_tracked of every BatchNorm (nn.BatchNorm2d.forward)
        model._stats_dirty = True         # the kernel updated the running statistics in the flat blob
        model._fwd_serial += 1
        ctx.model, ctx.serial = model, model._fwd_serial
        ctx.save_for_backward(spec)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        model = ctx.model
        if ctx.serial != model._fwd_serial:
            raise RuntimeError("backward of a train-mode forward that is no longer the most recent one: the HIP "
                               "trainer keeps the activations of one forward pass")
        (spec,) = ctx.saved_tensors
        # a fresh buffer per backward (p.grad may still alias the previous one under gradient accumulation); the
        # 248 returned gradients are views of it, so autograd leaves every p.grad pointing into ONE contiguous
        # tensor in the canonical layout -- train.allreduce_gradients() moves it as a single RCCL message
        grads = model._trainer(spec.device).backward(model._flat, spec, grad_out)
        model._grad_flat = grads
        return (None, None) + tuple(grads[o:o + n].view(shape) for o, n, shape in model._train_slices)


class GTCRNMicro(nn.Module):
    """Drop-in for ``gtcrn_micro.models.gtcrn_micro.GTCRNMicro`` on MI355X: eval-mode inference through
    the fused kernels, train-mode forward/backward through the layer-at-a-time training kernels."""

    def __init__(self, n_fft=512, hop_len=256, win_len=512):
        super().__init__()
        # architecture is fixed at 512/256/257 bins, exactly like the reference (args ignored)
        self.erb = ERB(65, 64)
        self.sfe = SFE_Lite(3)
        self.encoder = Encoder()
        self.gtcn1 = GTCN(16, n_layers=4, kernel_size=3, dilation=2)
        self.gtcn2 = GTCN(16, n_layers=4, kernel_size=3, dilation=2)
        self.decoder = Decoder()
        self.mask = Mask()
        self._engines = {}     # device index -> (Engine, version signature)
        self._trainers = {}    # device index -> Trainer
        self._flat = None      # canonical blob on the device; every parameter/buffer is a view into it
        self._nbt_flat = None
        self._stats_dirty = False
        self._fwd_serial = 0
        self._opt_serial = 0    # in-place weight updates by the fused optimizer kernel (utils/optim.py)
        self._train_slices = None
        self._train_params = None
        self._grad_flat = None  # gradient blob of the most recent backward (canonical layout)
        self._act_storage = "f32"
        self._sig_tensors = None
        self._sig_epoch = -1
        self._obj_serial = 0

    # -- weight hand-over -------------------------------------------------------------------
    def _state_tensors(self):
        """The tensors behind the state_dict, collected once (walking 388 keys on every call costs more host time
        than a streaming frame takes on the GPU); dropped whenever storages are re-created."""
        if self._sig_tensors is None or self._sig_epoch != _REG_EPOCH[0] or not _HAVE_REG_HOOKS:
            self._sig_epoch = _REG_EPOCH[0]
            sd = self.state_dict(keep_vars=True)
            new = ([sd[name] for name, _, _ in _lib.param_table()],
                   [v for k, v in sd.items() if k.endswith("num_batches_tracked")])
            old = self._sig_tensors
            if old is not None and not all(a is b for a, b in zip(old[0], new[0])):
                self._obj_serial += 1          # one of THIS model's tensors was replaced: engines must re-fold
            self._sig_tensors = new
        return self._sig_tensors

    def _signature(self):
        """Changes whenever a weight may have changed: in-place updates (optimiser steps, load_state_dict's copy_,
        manual edits) bump the tensors' version counters; the train forward (running statistics) bumps the serial; a
        REPLACED parameter / buffer object moves the registration epoch."""
        ts = self._state_tensors()[0]
        return (sum(int(t._version) for t in ts), self._fwd_serial, self._obj_serial, self._opt_serial)

    # -- train mode: flat parameter storage ----------------------------------------------------
    def _trainer(self, device):
        idx = torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
        if idx not in self._trainers:
            self._trainers[idx] = _lib.Trainer(idx)
            self._trainers[idx].set_storage(self._act_storage)
        return self._trainers[idx]

    def set_activation_storage(self, storage):
        """Train mode only: keep the saved activations in "f32" (default, the reference's precision), "bf16"
        (BASELINE configs[3]: halves the HBM traffic and the workspace of the layer-at-a-time train step; arithmetic,
        BatchNorm statistics, gradients, Adam and the weights stay fp32 -- but the forward then IS the bf16-activation
        network), "bf16_saves" (bf16 copies for the backward only: the forward chain stays fp32, so the output equals
        the "f32" mode's bit for bit and the gradient differs from it only by the rounding of the saved tensors, at the
        bf16 mode's workspace) or "bf16_grads" ("bf16" with the gradients handed from one layer's backward to the next
        stored in bf16 as well -- what bf16 autocast keeps; same forward as "bf16", the fastest step).  Takes effect at
        the next forward."""
        if storage not in _lib.Trainer.STORAGE:
            raise _lib.GtcrnError(f"storage must be one of {sorted(_lib.Trainer.STORAGE)}, got {storage!r}")
        self._act_storage = storage
        for tr in self._trainers.values():
            tr.set_storage(storage)

    def _flatten(self, device):
        """Moves the storage of every parameter and buffer into ONE canonical blob on ``device`` (the layout
        of the C ABI) and re-points the tensors at views of it: optimiser updates, ``load_state_dict`` and
        the kernels' running-statistics updates then all act on the same memory, with no per-step packing."""
        table = _lib.param_table()
        sd = self.state_dict(keep_vars=True)
        for name, _, _ in table:
            if sd[name].device != device:
                raise _lib.GtcrnError(f"parameter '{name}' lives on {sd[name].device}, the input on {device}: "
                                      "move the model with .to(device) first (there is no CPU path)")
        blob = torch.empty(_lib.NPARAM_FLOATS, device=device, dtype=torch.float32)
        slices = []
        with torch.no_grad():
            for name, numel, off in table:
                t = sd[name]
                blob[off:off + numel].copy_(t.reshape(-1))
                t.data = blob[off:off + numel].view(t.shape)
                if _is_trainable(name):
                    slices.append((off, numel, tuple(t.shape)))
            nbt = [v for k, v in sd.items() if k.endswith("num_batches_tracked")]
            flat_n = torch.stack([v.reshape(()) for v in nbt]).to(device)
            for i, v in enumerate(nbt):
                v.data = flat_n[i]
        self._flat, self._nbt_flat, self._train_slices = blob, flat_n, slices
        self._train_params = [sd[name] for name, _, _ in table if _is_trainable(name)]

    def _flat_ok(self, device):
        if self._flat is None or self._flat.device != device:
            return False
        base = self._flat.data_ptr()
        return all(t.data_ptr() == base + 4 * off
                   for t, (_, _, off) in zip(self._state_tensors()[0], _lib.param_table()))

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)     # .to()/.cuda()/.float() re-create the storages
        self._flat = None
        self._sig_tensors = None
        return out

    def engine(self, device):
        """The HIP model handle for ``device``, re-folded whenever a parameter changed."""
        idx = torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
        sig = self._signature()
        ent = self._engines.get(idx)
        if ent is not None and ent[1] == sig:
            return ent[0]
        # after training the parameters live in one flat device blob already in the C ABI's order: one copy
        dev = torch.device("cuda", idx)
        blob = self._flat.cpu().numpy() if self._flat_ok(dev) else state_dict_to_blob(self.state_dict())
        if ent is None:
            eng = _lib.Engine(blob, idx)
        else:
            eng = ent[0]
            eng.set_params(blob)
        self._engines[idx] = (eng, sig)
        return eng

    def load_state_dict(self, state_dict, strict=True, assign=False):
        if any(k.startswith("module.") for k in state_dict):
            state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._sig_tensors = None
        for idx, (eng, _) in list(self._engines.items()):
            self._engines[idx] = (eng, None)   # force a re-fold on next use
        return out

    def forward(self, spec):
        """spec: (B, 257, T, 2) float32 on the GPU -> enhanced (B, 257, T, 2)."""
        if not spec.is_cuda:
            raise _lib.GtcrnError("GTCRNMicro.forward needs a CUDA (ROCm) tensor: there is no CPU path")
        if self.training:
            # nn.BatchNorm2d in .train() mode: batch statistics + running-statistics update (train.py:265)
            if not self._flat_ok(spec.device):
                self._flatten(spec.device)
            if torch.is_grad_enabled():
                return _TrainStep.apply(spec, self, *self._train_params)
            return _TrainStep.forward(_NoCtx(), spec, self)
        # .eval(): the fused inference kernels have no backward.  The reference's module would build a graph here when
        # grad mode is on; a caller that asks for the INPUT gradient gets an error instead of a tensor that silently
        # has no grad_fn.  (Parameters that require grad do not trigger it: the reference's own causality test calls
        # the eval-mode model outside no_grad, tests/models/test_gtcrn_micro.py:7-24 -- INTEGRATION.md section 5.)
        if torch.is_grad_enabled() and spec.requires_grad:
            raise _lib.GtcrnError("GTCRNMicro.forward in .eval() mode cannot be differentiated (the fused inference "
                                  "kernels keep no activations): call .train() for a backward pass, or detach the "
                                  "input / use torch.no_grad() for inference")
        return self.engine(spec.device).forward_spec(spec)


class _NoCtx:
    """Stand-in for the autograd context when a train-mode forward runs under torch.no_grad()."""

    def save_for_backward(self, *a):
        pass
