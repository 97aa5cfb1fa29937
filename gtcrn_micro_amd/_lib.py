"""ctypes binding of include/gtcrn_micro_hip.h (the C-ABI drop-in boundary).

The product path is HIP only: if the shared library is missing, or no gfx950
device is present, everything here raises -- there is no CPU fallback and
nothing under oracle/ is ever imported from this package.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgtcrn_micro_hip.so")
NPARAM_FLOATS = 44938
NBINS = 257

_c_f32p = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p
_lib = None


class GtcrnError(RuntimeError):
    pass


def lib():
    """Loads libgtcrn_micro_hip.so (built in-tree by gtcrn_micro_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    variant = os.environ.get("GTCRN_LIB_VARIANT")
    if variant in ("stamps", "exp"):     # diagnostic builds: tools/phase_profile.py / tools/ab_bench.py only
        path = LIB_PATH.replace(".so", f"_{variant}.so")
    if not os.path.exists(path):
        raise GtcrnError(
            f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the HIP extension is mandatory, there is no CPU path)")
    # PyTorch ships its own libamdhip64; loading it first makes this library bind to the same runtime
    # (two HIP runtimes in one process: the second one finds no GPU)
    import torch  # noqa: F401
    L = ctypes.CDLL(path)
    ci, cl = ctypes.c_int, ctypes.c_long
    L.gtcrn_abi_version.restype = ci
    L.gtcrn_last_error.restype = ctypes.c_char_p
    L.gtcrn_param_tensors.restype = cl
    L.gtcrn_param_name.restype = ctypes.c_char_p
    L.gtcrn_param_name.argtypes = [cl]
    L.gtcrn_param_numel.restype = cl
    L.gtcrn_param_numel.argtypes = [cl]
    L.gtcrn_param_offset.restype = cl
    L.gtcrn_param_offset.argtypes = [cl]
    L.gtcrn_model_create.argtypes = [ctypes.POINTER(_vp), _c_f32p, cl, ci]
    L.gtcrn_model_set_params.argtypes = [_vp, _c_f32p, cl]
    L.gtcrn_model_destroy.argtypes = [_vp]
    L.gtcrn_model_destroy.restype = None
    L.gtcrn_model_reserve.argtypes = [_vp, ci, ci]
    L.gtcrn_make_window.argtypes = [ci, _c_f32p]
    L.gtcrn_num_frames.restype = cl
    L.gtcrn_num_frames.argtypes = [cl]
    L.gtcrn_stft.argtypes = [_vp, ci, cl, _vp, _vp, cl, cl, cl, _vp]
    L.gtcrn_stft_frames.argtypes = [_vp, ci, cl, _vp, _vp, _vp]
    L.gtcrn_istft.argtypes = [_vp, cl, cl, cl, ci, ci, _vp, _vp, _vp]
    L.gtcrn_forward_spec.argtypes = [_vp, _vp, cl, cl, cl, _vp, cl, cl, cl, ci, ci, _vp]
    L.gtcrn_forward_wave.argtypes = [_vp, _vp, _vp, ci, cl, _vp, _vp]
    L.gtcrn_forward_wave_var.argtypes = [_vp, _vp, _vp, ci, cl, _vp, _vp, _vp]
    cf = ctypes.c_float
    L.gtcrn_forward_spec_quant.argtypes = [_vp, _vp, cl, cl, cl, _vp, cl, cl, cl, ci, ci, cf, cf, _vp]
    L.gtcrn_forward_wave_quant.argtypes = [_vp, _vp, _vp, ci, cl, _vp, cf, cf, _vp]
    L.gtcrn_pack_params_quant_host.argtypes = [_c_f32p, cl, _c_f32p, ctypes.POINTER(ci)]
    L.gtcrn_round_to_half.restype = cf
    L.gtcrn_round_to_half.argtypes = [cf]
    L.gtcrn_stream_state_bytes.restype = ctypes.c_size_t
    L.gtcrn_stream_reset.argtypes = [_vp, _vp, ci, _vp]
    L.gtcrn_stream_step.argtypes = [_vp, _vp, _vp, cl, cl, cl, _vp, cl, cl, cl, ci, ci, _vp]
    L.gtcrn_stream_import.argtypes = [_vp, _vp, ci, _vp, _vp, ctypes.POINTER(_vp), _vp]
    L.gtcrn_stream_export.argtypes = [_vp, _vp, ci, _vp, _vp, ctypes.POINTER(_vp), _vp]
    L.gtcrn_stream_conv2d.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp] + [ci] * 12 + [_vp]
    L.gtcrn_pack_sizes.argtypes = [ctypes.POINTER(cl), ctypes.POINTER(cl)]
    L.gtcrn_pack_sizes.restype = None
    L.gtcrn_pack_params_host.argtypes = [_c_f32p, cl, _c_f32p, ctypes.POINTER(ci)]
    L.gtcrn_debug_enable.argtypes = [_vp, ci]
    L.gtcrn_var_spans_enable.argtypes = [_vp, ci]
    L.gtcrn_stream_form.argtypes = [_vp, ci]
    L.gtcrn_stream_streams_per_workgroup.argtypes = [ci]
    L.gtcrn_debug_tap.restype = cl
    L.gtcrn_debug_tap.argtypes = [_vp, ctypes.c_char_p, ci, _c_f32p, cl]
    L.gtcrn_debug_stamps.restype = cl
    L.gtcrn_debug_stamps.argtypes = [_vp, ci, ctypes.POINTER(ctypes.c_ulonglong), cl]
    L.gtcrn_selftest_mfma.argtypes = [ci]
    L.gtcrn_pcm16_to_f32.argtypes = [ci, _vp, _vp, cl, _vp]
    L.gtcrn_f32_to_pcm16.argtypes = [ci, _vp, _vp, cl, _vp]
    L.gtcrn_selftest_split3.argtypes = [ci, _c_f32p, cl, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p]
    L.gtcrn_timing_enable.argtypes = [_vp, ci]
    L.gtcrn_timing_read.argtypes = [_vp, ci, ctypes.c_char_p, ci, _c_f32p, ctypes.POINTER(ci)]
    L.gtcrn_timing_kernels.restype = ci
    L.gtcrn_trainer_create.argtypes = [ctypes.POINTER(_vp), ci]
    L.gtcrn_trainer_destroy.argtypes = [_vp]
    L.gtcrn_trainer_destroy.restype = None
    L.gtcrn_train_workspace_bytes.restype = cl
    L.gtcrn_train_workspace_bytes.argtypes = [ci, ci]
    L.gtcrn_train_workspace_bytes2.restype = cl
    L.gtcrn_train_workspace_bytes2.argtypes = [ci, ci, ci]
    L.gtcrn_trainer_workspace_bytes.restype = cl
    L.gtcrn_trainer_workspace_bytes.argtypes = [_vp, ci, ci]
    L.gtcrn_trainer_set_storage.argtypes = [_vp, ci]
    L.gtcrn_trainer_set_fusions.argtypes = [_vp, ci]
    L.gtcrn_train_forward.argtypes = [_vp, _vp, _vp, cl, cl, cl, _vp, cl, cl, cl, ci, ci, _vp]
    L.gtcrn_train_backward.argtypes = [_vp, _vp, _vp, cl, cl, cl, _vp, cl, cl, cl, _vp, _vp]
    L.gtcrn_train_tap.argtypes = [_vp, ctypes.c_char_p, _vp, ctypes.POINTER(cl), _vp]
    L.gtcrn_train_loss.argtypes = [_vp, _vp, cl, cl, cl, _vp, cl, cl, cl, ci, ci, _vp, _vp, _vp]
    L.gtcrn_train_loss_strided.argtypes = [_vp, _vp, cl, cl, cl, _vp, cl, cl, cl, ci, ci, _vp, _vp, cl, cl, cl, _vp]
    cd = ctypes.c_double
    L.gtcrn_clip_adam_step.argtypes = [ci, _vp, _vp, _vp, _vp, _vp, cl, cf, cd, cd, cd, cd, cd, cl, _vp, _vp, _vp]
    L.gtcrn_clip_adam_workspace_bytes.restype = cl
    L.gtcrn_clip_adam_workspace_bytes.argtypes = [cl]
    if L.gtcrn_abi_version() != 1:
        raise GtcrnError("libgtcrn_micro_hip.so ABI version mismatch")
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise GtcrnError(lib().gtcrn_last_error().decode() or f"gtcrn error {rc}")
    return rc


_param_table = None


def param_table():
    """((name, numel, offset), ...) of the canonical blob (reference state_dict order); built once."""
    global _param_table
    if _param_table is None:
        L = lib()
        _param_table = tuple((L.gtcrn_param_name(i).decode(), L.gtcrn_param_numel(i), L.gtcrn_param_offset(i))
                             for i in range(L.gtcrn_param_tensors()))
    return _param_table


def make_window(kind=0):
    w = np.empty(512, np.float32)
    _check(lib().gtcrn_make_window(kind, w.ctypes.data_as(_c_f32p)))
    return w


def num_frames(L):
    return int(lib().gtcrn_num_frames(int(L)))


def pack_params_host(params, quant=False):
    """Host-only: the BN-folded slot-space buffers (floats, ints) the kernels consume; quant=True: with every conv /
    linear weight as fp16(int8 * per-output-channel scale) (the configs[4] variant)."""
    params = np.ascontiguousarray(params, np.float32).ravel()
    nf, ni = ctypes.c_long(), ctypes.c_long()
    lib().gtcrn_pack_sizes(ctypes.byref(nf), ctypes.byref(ni))
    F = np.empty(nf.value, np.float32)
    I = np.empty(ni.value, np.int32)
    fn = lib().gtcrn_pack_params_quant_host if quant else lib().gtcrn_pack_params_host
    _check(fn(params.ctypes.data_as(_c_f32p), params.size, F.ctypes.data_as(_c_f32p),
              I.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
    return F, I


def round_to_half(x):
    """float -> IEEE binary16 (round to nearest even) -> float, the host twin of the kernels' v_cvt_f16_f32."""
    return float(lib().gtcrn_round_to_half(float(x)))


def _stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def _spec_strides(t):
    """(sb, sf, st) in elements of a (B,257,T,2) float32 tensor whose last dim is contiguous."""
    assert t.dim() == 4 and t.shape[1] == NBINS and t.shape[3] == 2
    if t.stride(3) != 1:
        raise GtcrnError("the re/im pair of a spectrogram must be contiguous")
    return t.stride(0), t.stride(1), t.stride(2)


def empty_spec(B, T, device, frame_major=False):
    """An uninitialised (B,257,T,2) float32 spectrogram.  frame_major: the memory is (B,T,257,2) -- a frame's 257 bins are
    one 2 KB row, which is how every kernel of this library walks a spectrogram -- viewed in the reference's shape."""
    import torch
    if frame_major:
        return torch.empty((B, T, NBINS, 2), device=device, dtype=torch.float32).permute(0, 2, 1, 3)
    return torch.empty((B, NBINS, T, 2), device=device, dtype=torch.float32)


def _empty_spec_like(t):
    """A new spectrogram with the shape of t and t's memory order (frame-major stays frame-major)."""
    B, _, T, _ = t.shape
    return empty_spec(B, T, t.device, frame_major=abs(t.stride(2)) > abs(t.stride(1)))


def _require_cuda_f32(t, what):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise GtcrnError(f"{what} must be a CUDA (ROCm) tensor: this implementation has no CPU path")
    if t.dtype != torch.float32:
        raise GtcrnError(f"{what} must be float32")


def stft(wave, window, out=None, frame_major=False):
    """torch.stft(x,512,256,512,window,return_complex=False) on the GPU: (B,L) or (L,) -> (B,257,T,2)/(257,T,2).
    frame_major: see empty_spec (same shape and values, the library's preferred memory order)."""
    import torch
    _require_cuda_f32(wave, "wave")
    squeeze = wave.dim() == 1
    w2 = wave.reshape(1, -1) if squeeze else wave
    w2 = w2.contiguous()
    B, L = w2.shape
    if L < 257:
        raise GtcrnError("reflect padding needs more than 256 samples")
    T = num_frames(L)
    win = window.to(device=wave.device, dtype=torch.float32).contiguous()
    spec = out if out is not None else empty_spec(B, T, wave.device, frame_major)
    sb, sf, st = _spec_strides(spec)
    with torch.cuda.device(wave.device):
        _check(lib().gtcrn_stft(w2.data_ptr(), B, L, win.data_ptr(), spec.data_ptr(), sb, sf, st, _stream_ptr()))
    return spec[0] if squeeze else spec


def stft_frames(wave, window):
    import torch
    _require_cuda_f32(wave, "wave")
    w2 = (wave.reshape(1, -1) if wave.dim() == 1 else wave).contiguous()
    B, L = w2.shape
    T = num_frames(L)
    win = window.to(device=wave.device, dtype=torch.float32).contiguous()
    fr = torch.empty((B, T, 512), device=wave.device, dtype=torch.float32)
    with torch.cuda.device(wave.device):
        _check(lib().gtcrn_stft_frames(w2.data_ptr(), B, L, win.data_ptr(), fr.data_ptr(), _stream_ptr()))
    return fr


def istft(spec, window):
    """torch.istft(view_as_complex(spec),512,256,512,window): (B,257,T,2)/(257,T,2) -> (B,256(T-1))/(256(T-1),)."""
    import torch
    _require_cuda_f32(spec, "spec")
    squeeze = spec.dim() == 3
    s4 = spec.unsqueeze(0) if squeeze else spec
    if s4.stride(3) != 1:
        s4 = s4.contiguous()
    B, _, T, _ = s4.shape
    if T < 2:
        raise GtcrnError("iSTFT needs at least 2 frames")
    win = window.to(device=spec.device, dtype=torch.float32).contiguous()
    out = torch.empty((B, 256 * (T - 1)), device=spec.device, dtype=torch.float32)
    sb, sf, st = _spec_strides(s4)
    with torch.cuda.device(spec.device):
        _check(lib().gtcrn_istft(s4.data_ptr(), sb, sf, st, B, T, win.data_ptr(), out.data_ptr(), _stream_ptr()))
    return out[0] if squeeze else out


def stream_conv2d(x, cache, weight, bias, kt, kf, dt=1, df=1, pad_f=0, groups=1, transposed=False):
    """Causal streaming conv step on the GPU (see gtcrn_stream_conv2d): returns (y, new_cache)."""
    import torch
    _require_cuda_f32(x, "x")
    x = x.contiguous()
    B, Cin, T, F = x.shape
    Cout = weight.shape[1] if transposed else weight.shape[0]
    H = (kt - 1) * dt
    if cache is None:
        cache = torch.zeros((B, Cin, H, F), device=x.device, dtype=torch.float32)
    cache = cache.contiguous()
    Fout = F - 2 * pad_f + df * (kf - 1) if transposed else F + 2 * pad_f - df * (kf - 1)
    y = torch.empty((B, Cout, T, Fout), device=x.device, dtype=torch.float32)
    new_cache = torch.empty_like(cache)
    w = weight.detach().to(device=x.device, dtype=torch.float32).contiguous()
    bptr = bias.detach().to(device=x.device, dtype=torch.float32).contiguous() if bias is not None else None
    with torch.cuda.device(x.device):
        rc = lib().gtcrn_stream_conv2d(x.data_ptr(), cache.data_ptr(), w.data_ptr(),
                                       bptr.data_ptr() if bptr is not None else None, y.data_ptr(),
                                       new_cache.data_ptr(), B, Cin, Cout, T, F, kt, kf, dt, df, pad_f, groups,
                                       int(bool(transposed)), _stream_ptr())
    _check(rc)
    assert rc == Fout
    return y, new_cache


class Engine:
    """One model handle (gtcrn_model) on one device."""

    def __init__(self, params, device=0):
        params = np.ascontiguousarray(params, np.float32).ravel()
        if params.size != NPARAM_FLOATS:
            raise GtcrnError(f"parameter blob must hold {NPARAM_FLOATS} floats, got {params.size}")
        self.device = int(device)
        h = ctypes.c_void_p()
        _check(lib().gtcrn_model_create(ctypes.byref(h), params.ctypes.data_as(_c_f32p), params.size, self.device))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().gtcrn_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, params):
        params = np.ascontiguousarray(params, np.float32).ravel()
        _check(lib().gtcrn_model_set_params(self._h, params.ctypes.data_as(_c_f32p), params.size))

    def reserve(self, B, T):
        _check(lib().gtcrn_model_reserve(self._h, int(B), int(T)))

    def _dev(self):
        import torch
        return torch.cuda.device(self.device)

    def forward_spec(self, spec, out=None):
        import torch
        _require_cuda_f32(spec, "spec")
        if spec.dim() != 4 or spec.shape[1] != NBINS or spec.shape[3] != 2:
            raise GtcrnError(f"spec must be (B,257,T,2), got {tuple(spec.shape)}")
        if spec.device.index != self.device:
            raise GtcrnError("spec is on a different device than the model")
        if spec.stride(3) != 1:
            spec = spec.contiguous()
        B, _, T, _ = spec.shape
        if out is None:
            out = torch.empty((B, NBINS, T, 2), device=spec.device, dtype=torch.float32)
        else:
            self._check_on_device(out, "out")
            if tuple(out.shape) != (B, NBINS, T, 2) or out.stride(3) != 1:
                raise GtcrnError(f"out must be (B,257,{T},2) with contiguous re/im pairs, got {tuple(out.shape)}")
        isb, isf, ist = _spec_strides(spec)
        osb, osf, ost = _spec_strides(out)
        with self._dev():
            _check(lib().gtcrn_forward_spec(self._h, spec.data_ptr(), isb, isf, ist, out.data_ptr(), osb, osf, ost,
                                            B, T, _stream_ptr()))
        return out

    def _check_on_device(self, t, what, shape=None):
        import torch
        _require_cuda_f32(t, what)
        if t.device.index != self.device:
            raise GtcrnError(f"{what} is on cuda:{t.device.index}, the model on cuda:{self.device}")
        if shape is not None and (tuple(t.shape) != tuple(shape) or not t.is_contiguous()):
            raise GtcrnError(f"{what} must be a contiguous float32 tensor of shape {tuple(shape)}, got "
                             f"{tuple(t.shape)} (contiguous: {t.is_contiguous()})")

    def forward_wave(self, wave, window, out=None):
        import torch
        self._check_on_device(wave, "wave")
        w2 = (wave.reshape(1, -1) if wave.dim() == 1 else wave).contiguous()
        if w2.dim() != 2:
            raise GtcrnError(f"wave must be (B,L) or (L,), got {tuple(wave.shape)}")
        B, L = w2.shape
        T = num_frames(L)
        win = window.to(device=wave.device, dtype=torch.float32).contiguous()
        if out is None:
            out = torch.empty((B, 256 * (T - 1)), device=wave.device, dtype=torch.float32)
        else:
            self._check_on_device(out, "out", (B, 256 * (T - 1)))
        with self._dev():
            _check(lib().gtcrn_forward_wave(self._h, w2.data_ptr(), out.data_ptr(), B, L, win.data_ptr(),
                                            _stream_ptr()))
        return out[0] if wave.dim() == 1 else out

    # ---- int8-weight / fp16-activation variant (BASELINE configs[4]; contract in include/gtcrn_micro_hip.h) ------
    def forward_spec_quant(self, spec, in_scale=0.0, out_scale=0.0, out=None):
        import torch
        self._check_on_device(spec, "spec")
        if spec.dim() != 4 or spec.shape[1] != NBINS or spec.shape[3] != 2:
            raise GtcrnError(f"spec must be (B,257,T,2), got {tuple(spec.shape)}")
        if spec.stride(3) != 1:
            spec = spec.contiguous()
        B, _, T, _ = spec.shape
        if out is None:
            out = torch.empty((B, NBINS, T, 2), device=spec.device, dtype=torch.float32)
        else:
            self._check_on_device(out, "out")
            if tuple(out.shape) != (B, NBINS, T, 2) or out.stride(3) != 1:
                raise GtcrnError(f"out must be (B,257,{T},2) with contiguous re/im pairs, got {tuple(out.shape)}")
        isb, isf, ist = _spec_strides(spec)
        osb, osf, ost = _spec_strides(out)
        with self._dev():
            _check(lib().gtcrn_forward_spec_quant(self._h, spec.data_ptr(), isb, isf, ist, out.data_ptr(), osb, osf,
                                                  ost, B, T, float(in_scale), float(out_scale), _stream_ptr()))
        return out

    def forward_wave_quant(self, wave, window, in_scale=0.0, out_scale=0.0, out=None):
        import torch
        self._check_on_device(wave, "wave")
        w2 = (wave.reshape(1, -1) if wave.dim() == 1 else wave).contiguous()
        if w2.dim() != 2:
            raise GtcrnError(f"wave must be (B,L) or (L,), got {tuple(wave.shape)}")
        B, L = w2.shape
        T = num_frames(L)
        win = window.to(device=wave.device, dtype=torch.float32).contiguous()
        if out is None:
            out = torch.empty((B, 256 * (T - 1)), device=wave.device, dtype=torch.float32)
        else:
            self._check_on_device(out, "out", (B, 256 * (T - 1)))
        with self._dev():
            _check(lib().gtcrn_forward_wave_quant(self._h, w2.data_ptr(), out.data_ptr(), B, L, win.data_ptr(),
                                                  float(in_scale), float(out_scale), _stream_ptr()))
        return out[0] if wave.dim() == 1 else out

    def forward_wave_var(self, wave, lengths, window, out=None):
        """Clips of different lengths through ONE launch sequence: ``wave`` (B,Lmax) holds clip b in its first
        ``lengths[b]`` samples (257 <= lengths[b] <= Lmax).  Returns (B, 256*(Lmax//256)); row b carries its
        256*(lengths[b]//256) enhanced samples (bit-identical to forward_wave on that clip alone), the rest of the
        row is unspecified."""
        import torch
        self._check_on_device(wave, "wave")
        if wave.dim() != 2 or not wave.is_contiguous():
            raise GtcrnError("wave must be a contiguous (B,Lmax) tensor")
        B, L = wave.shape
        lens = torch.as_tensor(lengths, dtype=torch.int32).reshape(-1)
        if lens.numel() != B:
            raise GtcrnError(f"lengths must hold {B} entries, got {lens.numel()}")
        lmin, lmax = int(lens.min()), int(lens.max())
        if lmin < 257 or lmax > L:
            raise GtcrnError(f"every length must lie in [257, Lmax={L}], got min {lmin} max {lmax}")
        lens = lens.to(wave.device)
        T = num_frames(L)
        win = window.to(device=wave.device, dtype=torch.float32).contiguous()
        if out is None:
            out = torch.empty((B, 256 * (T - 1)), device=wave.device, dtype=torch.float32)
        else:
            self._check_on_device(out, "out", (B, 256 * (T - 1)))
        with self._dev():
            _check(lib().gtcrn_forward_wave_var(self._h, wave.data_ptr(), out.data_ptr(), B, L, lens.data_ptr(),
                                                win.data_ptr(), _stream_ptr()))
        return out

    # ---- streaming -----------------------------------------------------------------------
    @staticmethod
    def state_bytes():
        return int(lib().gtcrn_stream_state_bytes())

    def new_state(self, nstreams):
        import torch
        st = torch.empty((nstreams, self.state_bytes() // 4), device=f"cuda:{self.device}", dtype=torch.float32)
        with self._dev():
            _check(lib().gtcrn_stream_reset(self._h, st.data_ptr(), nstreams, _stream_ptr()))
        return st

    def stream_step(self, state, spec_t, out=None):
        import torch
        self._check_on_device(spec_t, "spec")
        if spec_t.dim() != 4 or spec_t.shape[1] != NBINS or spec_t.shape[3] != 2:
            raise GtcrnError(f"spec must be (N,257,n,2), got {tuple(spec_t.shape)}")
        if spec_t.stride(3) != 1:
            spec_t = spec_t.contiguous()
        N, _, nfr, _ = spec_t.shape
        self._check_on_device(state, "state", (N, self.state_bytes() // 4))
        if out is None:
            out = torch.empty((N, NBINS, nfr, 2), device=spec_t.device, dtype=torch.float32)
        else:
            self._check_on_device(out, "out")
            if tuple(out.shape) != (N, NBINS, nfr, 2) or out.stride(3) != 1:
                raise GtcrnError(f"out must be (N,257,{nfr},2) with contiguous re/im pairs, got {tuple(out.shape)}")
        isb, isf, ist = _spec_strides(spec_t)
        osb, osf, ost = _spec_strides(out)
        with self._dev():
            _check(lib().gtcrn_stream_step(self._h, state.data_ptr(), spec_t.data_ptr(), isb, isf, ist,
                                           out.data_ptr(), osb, osf, ost, N, nfr, _stream_ptr()))
        return out

    def _cache_ptrs(self, tcn_cache):
        flat = [tcn_cache[g][k] for g in range(2) for k in range(4)]
        for g in range(2):
            for k in range(4):
                _require_cuda_f32(tcn_cache[g][k], "tcn_cache")
                assert tcn_cache[g][k].is_contiguous()
        arr = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in flat])
        return arr

    def stream_import(self, state, conv_cache, tra_cache, tcn_cache):
        N = state.shape[0]
        assert conv_cache.is_contiguous() and tra_cache.is_contiguous()
        with self._dev():
            _check(lib().gtcrn_stream_import(self._h, state.data_ptr(), N, conv_cache.data_ptr(),
                                             tra_cache.data_ptr(), self._cache_ptrs(tcn_cache), _stream_ptr()))

    def stream_export(self, state, conv_cache, tra_cache, tcn_cache):
        N = state.shape[0]
        assert conv_cache.is_contiguous() and tra_cache.is_contiguous()
        with self._dev():
            _check(lib().gtcrn_stream_export(self._h, state.data_ptr(), N, conv_cache.data_ptr(),
                                             tra_cache.data_ptr(), self._cache_ptrs(tcn_cache), _stream_ptr()))

    # ---- test / measurement hooks ------------------------------------------------------------
    def debug_enable(self, on=True):
        """True / 1: stage taps (and phase stamps in the diagnostic build); 2: phase stamps only (single-frame streaming
        steps keep their one-launch form)."""
        _check(lib().gtcrn_debug_enable(self._h, 2 if on == 2 else int(bool(on))))

    def var_spans_enable(self, on=True):
        """Variable-length batches in time spans (default on; results are bit-identical either way -- the A/B switch)."""
        _check(lib().gtcrn_var_spans_enable(self._h, int(bool(on))))

    def stream_form(self, form=0):
        """Single-frame streaming steps: 0 = one launch per step, four or seven streams per workgroup by the stream count
        (default); 1 = the three-launch form; 2 / 3 = one launch pinned to four / seven streams per workgroup (A/B
        switches; bit-identical results)."""
        _check(lib().gtcrn_stream_form(self._h, int(form)))

    def tap(self, name, b, T):
        F = {"en0": 65, "de3": 65}.get(name, 33)
        shape = (2, T, 129) if name == "de4" else (16, T, F)
        dst = np.empty(shape, np.float32)
        n = lib().gtcrn_debug_tap(self._h, name.encode(), int(b), dst.ctypes.data_as(_c_f32p), dst.size)
        _check(n)
        assert n == dst.size, (n, dst.size)
        return dst

    def stamps(self, kernel, B):
        """Diagnostic build: (B,16) phase cycle sums of kernel 0 encoder, 1 gtcn1, 2 gtcn2, 3 decoder."""
        dst = np.zeros((B, 16), np.uint64)
        _check(lib().gtcrn_debug_stamps(self._h, int(kernel), dst.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)),
                                        dst.size))
        return dst

    _kernels = None

    @classmethod
    def kernel_names(cls):
        """The kernels the library times, in its own order (every timed launch records which one it was)."""
        if cls._kernels is None:
            buf = ctypes.create_string_buffer(64)
            ms, n = ctypes.c_float(), ctypes.c_int()
            names = []
            for i in range(lib().gtcrn_timing_kernels()):
                _check(lib().gtcrn_timing_read(None, -1 - i, buf, 64, ctypes.byref(ms), ctypes.byref(n)))
                names.append(buf.value.decode())
            cls._kernels = tuple(names)
        return cls._kernels

    def timing_enable(self, on=True, only=None):
        """HIP-event timing of the kernel launches; only="k_decoder" keeps the events of that kernel alone (an
        event pair costs a few microseconds of dispatch gap per launch)."""
        mode = 0 if not on else (1 if only is None else 2 + self.kernel_names().index(only))
        _check(lib().gtcrn_timing_enable(self._h, mode))

    def timing_read(self):
        """{kernel name: (average ms, launches)} over the launches recorded since timing_enable(True)."""
        out = {}
        buf = ctypes.create_string_buffer(64)
        ms, n = ctypes.c_float(), ctypes.c_int()
        for i in range(len(self.kernel_names())):
            _check(lib().gtcrn_timing_read(self._h, i, buf, 64, ctypes.byref(ms), ctypes.byref(n)))
            if n.value:
                out[buf.value.decode()] = (float(ms.value), int(n.value))
        return out


def _pcm_pair(pcm, wave, who):
    import torch
    for t, dt, what in ((pcm, torch.int16, "int16"), (wave, torch.float32, "float32")):
        if not (t.is_cuda and t.dtype == dt and t.is_contiguous()):
            raise GtcrnError(f"{who}: a contiguous {what} CUDA tensor is required")
    if pcm.numel() != wave.numel() or pcm.device != wave.device:
        raise GtcrnError(f"{who}: the two tensors must hold the same number of samples on one device")
    if pcm.numel() % 8 or pcm.data_ptr() % 16 or wave.data_ptr() % 16:
        raise GtcrnError(f"{who}: 16-byte aligned tensors, a sample count that is a multiple of 8")


def pcm16_to_f32(pcm, out=None):
    """int16 samples -> float32 waveform on the device (x = s / 32768, exact: what soundfile.read returns, infer.py:54);
    asynchronous on the current stream of the tensor's device (gtcrn_pcm16_to_f32)."""
    import torch
    if out is None:
        out = torch.empty(pcm.shape, dtype=torch.float32, device=pcm.device)
    _pcm_pair(pcm, out, "pcm16_to_f32")
    with torch.cuda.device(pcm.device):
        _check(lib().gtcrn_pcm16_to_f32(pcm.device.index, ctypes.c_void_p(pcm.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                        int(pcm.numel()), _stream_ptr()))
    return out


def f32_to_pcm16(wave, out=None):
    """float32 waveform -> int16 samples on the device: clip(rint(y * 32768), -32768, 32767), round half to even (the
    16-bit PCM sf.write / scipy.io.wavfile.write of np.rint produce, infer.py:113); asynchronous on the current stream of
    the tensor's device (gtcrn_f32_to_pcm16)."""
    import torch
    if out is None:
        out = torch.empty(wave.shape, dtype=torch.int16, device=wave.device)
    _pcm_pair(out, wave, "f32_to_pcm16")
    with torch.cuda.device(wave.device):
        _check(lib().gtcrn_f32_to_pcm16(wave.device.index, ctypes.c_void_p(wave.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                        int(wave.numel()), _stream_ptr()))
    return out


def stream_streams_per_workgroup(nstreams):
    """Streams per workgroup (4: k_stream_ms, 7: k_stream_wide) of the one-launch step the library picks for `nstreams`
    single-frame steps (host logic only)."""
    return _check(lib().gtcrn_stream_streams_per_workgroup(int(nstreams)))


def selftest_mfma(device=0):
    _check(lib().gtcrn_selftest_mfma(int(device)))


def selftest_split3(x, A=None, B=None, device=0):
    """Device-side split3 / join3 / split_mm6 (the dense 3x3's exact bf16 split) on the given fp32 values: returns
    (planes (3,n), joined (n,)) and, with A (16,32) and B (32,16), also D (16,16) = A @ B through the six products."""
    x = np.ascontiguousarray(x, np.float32).ravel()
    planes = np.empty((3, x.size), np.float32)
    joined = np.empty(x.size, np.float32)
    f = lambda a: a.ctypes.data_as(_c_f32p)
    if A is None:
        _check(lib().gtcrn_selftest_split3(int(device), f(x), x.size, f(planes), f(joined), None, None, None))
        return planes, joined
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    assert A.shape == (16, 32) and B.shape == (32, 16)
    D = np.empty((16, 16), np.float32)
    _check(lib().gtcrn_selftest_split3(int(device), f(x), x.size, f(planes), f(joined), f(A), f(B), f(D)))
    return planes, joined, D


class Trainer:
    """Train-mode forward/backward of the model on one device (gtcrn_trainer): batch-statistics
    BatchNorm, saved activations, gradients of the 248 trainable tensors in the canonical blob layout."""

    def __init__(self, device=0):
        self.device = int(device)
        h = ctypes.c_void_p()
        _check(lib().gtcrn_trainer_create(ctypes.byref(h), self.device))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().gtcrn_trainer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # "bf16_saves": what the backward re-reads is stored in bf16 (as in "bf16") while the forward chain itself stays fp32
    # -- the forward is the fp32 network's, bit for bit; the gradient differs only by the rounding of the saved tensors
    # "bf16_grads": "bf16" with the gradients handed between units stored in bf16 too (what bf16 autocast training keeps)
    STORAGE = {"f32": 0, "fp32": 0, "bf16": 1, "bf16_saves": 4, "bf16_grads": 5}

    @staticmethod
    def workspace_bytes(B, T, storage="f32"):
        return int(lib().gtcrn_train_workspace_bytes2(int(B), int(T), Trainer.STORAGE[storage]))

    def set_storage(self, storage):
        """Storage of the saved activations: "f32" (the reference's precision), "bf16" (BASELINE configs[3]: bf16
        activations in the forward too), "bf16_saves" (bf16 copies for the backward only, fp32 forward chain) or
        "bf16_grads" ("bf16" whose inter-unit gradient tensors are bf16 as well); arithmetic, statistics, parameter
        gradients and weights are fp32 in all of them."""
        if storage not in self.STORAGE:
            raise GtcrnError(f"storage must be one of {sorted(self.STORAGE)}, got {storage!r}")
        _check(lib().gtcrn_trainer_set_storage(self._h, self.STORAGE[storage]))
        self.storage = storage

    def planned_workspace_bytes(self, B, T):
        """Workspace of a (B, T) problem with THIS trainer's storage mode and fusion mask (the static
        ``workspace_bytes`` describes the default mask only)."""
        return int(_check(lib().gtcrn_trainer_workspace_bytes(self._h, int(B), int(T))))

    def set_fusions(self, mask):
        """Diagnostic (gtcrn_trainer_set_fusions): 65535 = all pass fusions (default), 32767 = the weight-gradient finishes launched one by one, 16383 = the TCN's dilated depthwise forward through the general kernel, 8191 = the pointwise forward convs through the general conv kernel too, 4095 = without point_bn2 on load, 2047 =
        also without the producer-written decoder sums, 1023 = also without the in-launch finish of the BatchNorm reductions (round 4's), 7 = round 3's (every activation stored, separate skip-gradient adds),
        0 = the layer-at-a-time passes."""
        _check(lib().gtcrn_trainer_set_fusions(self._h, int(mask)))

    def _check_blob(self, blob, what):
        _require_cuda_f32(blob, what)
        if blob.numel() != NPARAM_FLOATS or not blob.is_contiguous():
            raise GtcrnError(f"{what} must be a contiguous tensor of {NPARAM_FLOATS} floats")

    def forward(self, params, spec, out=None):
        """params: canonical blob on the device (running statistics are updated in place); spec (B,257,T,2)."""
        import torch
        self._check_blob(params, "params")
        _require_cuda_f32(spec, "spec")
        if spec.dim() != 4 or spec.shape[1] != NBINS or spec.shape[3] != 2:
            raise GtcrnError(f"spec must be (B,257,T,2), got {tuple(spec.shape)}")
        if spec.stride(3) != 1:
            spec = spec.contiguous()
        B, _, T, _ = spec.shape
        if out is None:
            out = _empty_spec_like(spec)
        isb, isf, ist = _spec_strides(spec)
        osb, osf, ost = _spec_strides(out)
        with torch.cuda.device(self.device):
            _check(lib().gtcrn_train_forward(self._h, params.data_ptr(), spec.data_ptr(), isb, isf, ist,
                                             out.data_ptr(), osb, osf, ost, B, T, _stream_ptr()))
        return out

    def backward(self, params, spec, grad_out, grads=None):
        """Gradients (canonical blob layout) of the most recent forward for the upstream gradient grad_out."""
        import torch
        self._check_blob(params, "params")
        _require_cuda_f32(grad_out, "grad_out")
        if spec.stride(3) != 1:
            spec = spec.contiguous()
        if grad_out.stride(3) != 1:
            grad_out = grad_out.contiguous()
        if grads is None:
            grads = torch.empty(NPARAM_FLOATS, device=params.device, dtype=torch.float32)
        self._check_blob(grads, "grads")
        isb, isf, ist = _spec_strides(spec)
        gsb, gsf, gst = _spec_strides(grad_out)
        with torch.cuda.device(self.device):
            _check(lib().gtcrn_train_backward(self._h, params.data_ptr(), spec.data_ptr(), isb, isf, ist,
                                              grad_out.data_ptr(), gsb, gsf, gst, grads.data_ptr(), _stream_ptr()))
        return grads

    def hybrid_loss(self, pred, true, want_grad=True):
        """HybridLoss (loss.py:30-71) of two (B,257,T,2) spectrograms: returns (loss 0-d tensor, d loss/d pred or None)."""
        import torch
        _require_cuda_f32(pred, "pred")
        _require_cuda_f32(true, "true")
        if pred.shape != true.shape or pred.dim() != 4 or pred.shape[1] != NBINS or pred.shape[3] != 2:
            raise GtcrnError(f"pred/true must both be (B,257,T,2), got {tuple(pred.shape)} and {tuple(true.shape)}")
        if pred.stride(3) != 1:
            pred = pred.contiguous()
        if true.stride(3) != 1:
            true = true.contiguous()
        B, _, T, _ = pred.shape
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        grad = _empty_spec_like(pred) if want_grad else None      # (the gradient takes pred's memory order)
        psb, psf, pst = _spec_strides(pred)
        tsb, tsf, tst = _spec_strides(true)
        gsb, gsf, gst = _spec_strides(grad) if want_grad else (0, 0, 0)
        with torch.cuda.device(self.device):
            _check(lib().gtcrn_train_loss_strided(self._h, pred.data_ptr(), psb, psf, pst, true.data_ptr(), tsb, tsf, tst,
                                                  B, T, loss.data_ptr(), grad.data_ptr() if want_grad else None,
                                                  gsb, gsf, gst, _stream_ptr()))
        return loss, grad

    def tap(self, name):
        """Train-mode activation of the most recent forward at a stage boundary, as (B,C,T,F)."""
        import torch
        sh = (ctypes.c_long * 4)()
        _check(lib().gtcrn_train_tap(self._h, name.encode(), None, sh, None))
        out = torch.empty(tuple(sh), device=f"cuda:{self.device}", dtype=torch.float32)
        with torch.cuda.device(self.device):
            _check(lib().gtcrn_train_tap(self._h, name.encode(), out.data_ptr(), sh, _stream_ptr()))
        return out.permute(0, 3, 1, 2).contiguous()


_adam_ws = {}


def clip_adam_workspace(device, n=None):
    """A zeroed workspace for clip_adam_step (ticket + partial sums; every call leaves the ticket at zero again).  One per
    caller that may step concurrently with another on the same device (FlatAdam owns one per instance)."""
    import torch
    n = NPARAM_FLOATS if n is None else n
    return torch.zeros((int(lib().gtcrn_clip_adam_workspace_bytes(n)) + 7) // 8, dtype=torch.float64, device=device)


def clip_adam_step(params, grads, exp_avg, exp_avg_sq, mask, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                   max_norm=0.0, norm_out=None, ws=None):
    """clip_grad_norm_(max_norm) + torch.optim.Adam.step() over flat fp32 device blobs in two launches
    (gtcrn_clip_adam_step).  norm_out: optional 2-float device tensor (total norm, clip coefficient).  ws: the caller's own
    workspace (clip_adam_workspace); without one a per-(device, size) workspace shared by such callers is used -- fine for
    one stream per device."""
    import torch
    n = params.numel()
    for t, what in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (mask, "mask")):
        _require_cuda_f32(t, what)
        if t.numel() != n or not t.is_contiguous() or t.device != params.device:
            raise GtcrnError(f"{what} must be a contiguous float32 tensor of {n} elements on {params.device}")
    if norm_out is not None:
        _require_cuda_f32(norm_out, "norm_out")
        if norm_out.numel() < 2 or not norm_out.is_contiguous():
            raise GtcrnError("norm_out must hold 2 contiguous floats")
    if ws is None:
        key = (params.device.index, n)
        ws = _adam_ws.get(key)
        if ws is None:      # zeroed once: every call leaves the ticket at zero again
            ws = _adam_ws[key] = clip_adam_workspace(params.device, n)
    elif ws.device != params.device or ws.numel() * ws.element_size() < int(lib().gtcrn_clip_adam_workspace_bytes(n)):
        raise GtcrnError("clip_adam_step: the workspace is on another device or too small")
    with torch.cuda.device(params.device):
        _check(lib().gtcrn_clip_adam_step(params.device.index, params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(),
                                          exp_avg_sq.data_ptr(), mask.data_ptr(), n, float(max_norm), float(lr),
                                          float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
                                          norm_out.data_ptr() if norm_out is not None else None, ws.data_ptr(),
                                          _stream_ptr()))
