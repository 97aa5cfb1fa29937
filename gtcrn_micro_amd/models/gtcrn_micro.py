"""Host-side mirror of the reference's ``gtcrn_micro.models.gtcrn_micro`` for MI355X.

``GTCRNMicro`` keeps the reference's public surface -- constructor arguments
(accepted and ignored, models/gtcrn_micro.py:486-504), ``forward(spec)`` on a
``(B,257,T,2)`` spectrogram (:506-532), the 391 ``state_dict`` keys of the shipped
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


class GTCRNMicro(nn.Module):
    """Drop-in for ``gtcrn_micro.models.gtcrn_micro.GTCRNMicro`` (eval-mode inference on MI355X)."""

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

    # -- weight hand-over -------------------------------------------------------------------
    def _signature(self):
        return sum(int(t._version) for t in self.state_dict(keep_vars=True).values())

    def engine(self, device):
        """The HIP model handle for ``device``, re-folded whenever a parameter changed."""
        idx = torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
        sig = self._signature()
        ent = self._engines.get(idx)
        if ent is None:
            eng = _lib.Engine(state_dict_to_blob(self.state_dict()), idx)
            self._engines[idx] = (eng, sig)
            return eng
        eng, old = ent
        if old != sig:
            eng.set_params(state_dict_to_blob(self.state_dict()))
            self._engines[idx] = (eng, sig)
        return eng

    def load_state_dict(self, state_dict, strict=True, assign=False):
        if any(k.startswith("module.") for k in state_dict):
            state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        for idx, (eng, _) in list(self._engines.items()):
            self._engines[idx] = (eng, None)   # force a re-fold on next use
        return out

    def forward(self, spec):
        """spec: (B, 257, T, 2) float32 on the GPU -> enhanced (B, 257, T, 2)."""
        if self.training:
            raise NotImplementedError(
                "train-mode forward/backward is not built yet (SURVEY.md section 8f row 2); call .eval()")
        if not spec.is_cuda:
            raise _lib.GtcrnError("GTCRNMicro.forward needs a CUDA (ROCm) tensor: there is no CPU path")
        return self.engine(spec.device).forward_spec(spec)
