"""ctypes front-end of the CPU oracle (oracle/gtcrn_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (gtcrn_micro_amd) never
imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgtcrn_oracle.so")
NPARAMS = 44938
NBINS = 257
F32P = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    src = os.path.join(_HERE, "gtcrn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libgtcrn_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class State(ctypes.Structure):
    """One stream's caches in the reference layout (gtcrn_micro_stream.py:618-623)."""
    _fields_ = [("conv", ctypes.c_float * (2 * 16 * 6 * 33)),
                ("tra", ctypes.c_float * (2 * 3 * 8 * 2)),
                ("tcn", ctypes.c_float * (2 * 16 * 30 * 33))]


def _p(a):
    return a.ctypes.data_as(F32P)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.gtcrn_oracle_create.restype = ctypes.c_void_p
        L.gtcrn_oracle_create.argtypes = [F32P, ctypes.c_long]
        L.gtcrn_oracle_destroy.argtypes = [ctypes.c_void_p]
        L.gtcrn_oracle_window.argtypes = [ctypes.c_int, F32P]
        L.gtcrn_oracle_num_frames.restype = ctypes.c_long
        L.gtcrn_oracle_num_frames.argtypes = [ctypes.c_long]
        L.gtcrn_oracle_stft.argtypes = [F32P, ctypes.c_int, ctypes.c_long, F32P, F32P]
        L.gtcrn_oracle_frames.argtypes = [F32P, ctypes.c_int, ctypes.c_long, F32P, F32P]
        L.gtcrn_oracle_istft.argtypes = [F32P, ctypes.c_int, ctypes.c_int, F32P, F32P]
        L.gtcrn_oracle_forward.argtypes = [ctypes.c_void_p, F32P, ctypes.c_int, ctypes.c_int, F32P,
                                           ctypes.c_void_p]
        L.gtcrn_oracle_tap.restype = ctypes.c_long
        L.gtcrn_oracle_tap.argtypes = [ctypes.c_void_p, ctypes.c_char_p, F32P]
        L.gtcrn_oracle_enhance.argtypes = [ctypes.c_void_p, F32P, ctypes.c_int, ctypes.c_long, ctypes.c_int, F32P]
        ci = ctypes.c_int
        L.gtcrn_oracle_conv2d_causal.argtypes = [F32P, F32P, ci, ci, ci, F32P, F32P, ci, ci, ci, ci, ci, ci, ci,
                                                 F32P, ctypes.POINTER(ci)]
        L.gtcrn_oracle_convT2d_causal.argtypes = [F32P, F32P, ci, ci, ci, F32P, F32P, ci, ci, ci, ci, ci, ci,
                                                  F32P, ctypes.POINTER(ci)]
        _lib = L
    return _lib


def window(kind=0):
    w = np.empty(512, np.float32)
    lib().gtcrn_oracle_window(kind, _p(w))
    return w


def num_frames(L):
    return int(lib().gtcrn_oracle_num_frames(L))


def stft(wave, win):
    wave = _f32(np.atleast_2d(wave))
    B, L = wave.shape
    T = num_frames(L)
    spec = np.empty((B, NBINS, T, 2), np.float32)
    rc = lib().gtcrn_oracle_stft(_p(wave), B, L, _p(_f32(win)), _p(spec))
    if rc:
        raise ValueError("oracle stft failed (need L > 256)")
    return spec


def frames(wave, win):
    wave = _f32(np.atleast_2d(wave))
    B, L = wave.shape
    T = num_frames(L)
    out = np.empty((B, T, 512), np.float32)
    if lib().gtcrn_oracle_frames(_p(wave), B, L, _p(_f32(win)), _p(out)):
        raise ValueError("oracle frames failed")
    return out


def istft(spec, win):
    spec = _f32(spec)
    B, F, T, _ = spec.shape
    assert F == NBINS
    out = np.empty((B, 256 * (T - 1)), np.float32)
    if lib().gtcrn_oracle_istft(_p(spec), B, T, _p(_f32(win)), _p(out)):
        raise ValueError("oracle istft failed (need T >= 2)")
    return out


class Oracle:
    def __init__(self, params):
        params = _f32(params).ravel()
        assert params.size == NPARAMS, params.size
        self._h = lib().gtcrn_oracle_create(_p(params), params.size)
        if not self._h:
            raise ValueError("oracle create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gtcrn_oracle_destroy(self._h)
            self._h = None

    def forward(self, spec, states=None):
        """(B,257,T,2)->(B,257,T,2); states: ctypes array of State (len B) or None."""
        spec = _f32(spec)
        B, F, T, _ = spec.shape
        assert F == NBINS
        out = np.empty_like(spec)
        sp = ctypes.cast(states, ctypes.c_void_p) if states is not None else None
        if lib().gtcrn_oracle_forward(self._h, _p(spec), B, T, _p(out), sp):
            raise ValueError("oracle forward failed")
        return out

    def tap(self, name, shape=None):
        n = lib().gtcrn_oracle_tap(self._h, name.encode(), None)
        if n == 0:
            raise KeyError(name)
        a = np.empty(n, np.float32)
        lib().gtcrn_oracle_tap(self._h, name.encode(), _p(a))
        return a.reshape(shape) if shape is not None else a

    def enhance(self, wave, window_kind=0):
        wave = _f32(np.atleast_2d(wave))
        B, L = wave.shape
        T = num_frames(L)
        out = np.empty((B, 256 * (T - 1)), np.float32)
        if lib().gtcrn_oracle_enhance(self._h, _p(wave), B, L, window_kind, _p(out)):
            raise ValueError("oracle enhance failed")
        return out


def new_states(B):
    return (State * B)()


def state_views(st):
    """numpy views of one State in the reference cache shapes."""
    conv = np.ctypeslib.as_array(st.conv).reshape(2, 16, 6, 33)
    tra = np.ctypeslib.as_array(st.tra).reshape(2, 3, 8, 2)
    tcn_flat = np.ctypeslib.as_array(st.tcn).reshape(2, 16 * 30 * 33)
    tcn = [[tcn_flat[g, 16 * 33 * 2 * (d - 1):16 * 33 * 2 * (d - 1) + 16 * 2 * d * 33].reshape(16, 2 * d, 33)
            for d in (1, 2, 4, 8)] for g in range(2)]
    return conv, tra, tcn


def conv2d_causal(x, hist, w, b, dt=1, df=1, pf=0, groups=1):
    x, w = _f32(x), _f32(w)
    Cin, T, F = x.shape
    Cout, _, kt, kf = w.shape
    Fout = F + 2 * pf - df * (kf - 1)
    y = np.empty((Cout, T, Fout), np.float32)
    fo = ctypes.c_int()
    lib().gtcrn_oracle_conv2d_causal(_p(x), _p(_f32(hist)) if hist is not None else None, Cin, T, F, _p(w),
                                     _p(_f32(b)) if b is not None else None, Cout, kt, kf, dt, df, pf, groups,
                                     _p(y), ctypes.byref(fo))
    assert fo.value == Fout
    return y


def convT2d_causal(x, hist, w, b, dt=1, df=1, pf=0):
    x, w = _f32(x), _f32(w)
    Cin, T, F = x.shape
    _, Cout, kt, kf = w.shape
    Fout = F - 2 * pf + df * (kf - 1)
    y = np.empty((Cout, T, Fout), np.float32)
    fo = ctypes.c_int()
    lib().gtcrn_oracle_convT2d_causal(_p(x), _p(_f32(hist)) if hist is not None else None, Cin, T, F, _p(w),
                                      _p(_f32(b)) if b is not None else None, Cout, kt, kf, dt, df, pf,
                                      _p(y), ctypes.byref(fo))
    assert fo.value == Fout
    return y
