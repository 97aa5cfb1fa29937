"""Host-side mirror of ``gtcrn_micro.streaming.gtcrn_micro_stream.StreamGTCRNMicro``.

``forward(spec, conv_cache, tra_cache, tcn_cache)`` keeps the reference call
(gtcrn_micro_stream.py:541-574): the caller owns the three caches in the reference
shapes -- conv_cache (2,B,16,6,33), tra_cache (2,3,B,8,2), tcn_cache 2 x 4 tensors
(B,16,2d,33) (:618-623) -- and gets them back updated (in place, like the reference).
Internally the HIP kernels keep per-stream ring state; ``forward`` imports the
caches, steps, and exports them again.  ``init_state`` / ``step`` is the native
form that leaves the state on the device between frames (no conversion per frame).
"""
import torch
import torch.nn as nn

from .. import _lib
from ..models.gtcrn_micro import GTCRNMicro


class StreamGTCRNMicro(GTCRNMicro):
    """Streaming twin: same parameters as the offline model (``convert_to_stream`` copies them)."""

    def __init__(self, n_fft=512, hop_len=256, win_len=512):
        super().__init__(n_fft, hop_len, win_len)
        self._scratch = {}

    @staticmethod
    def init_caches(batch, device):
        """Zero caches in the reference layout (gtcrn_micro_stream.py:618-623)."""
        conv_cache = torch.zeros(2, batch, 16, 6, 33, device=device)
        tra_cache = torch.zeros(2, 3, batch, 8, 2, device=device)
        tcn_cache = [[torch.zeros(batch, 16, 2 * d, 33, device=device) for d in (1, 2, 4, 8)] for _ in range(2)]
        return conv_cache, tra_cache, tcn_cache

    def forward(self, spec, conv_cache, tra_cache, tcn_cache):
        if self.training:
            raise NotImplementedError("streaming inference is eval-mode only; call .eval()")
        if not spec.is_cuda:
            raise _lib.GtcrnError("StreamGTCRNMicro.forward needs CUDA (ROCm) tensors: there is no CPU path")
        B = spec.shape[0]
        if tuple(conv_cache.shape) != (2, B, 16, 6, 33):
            raise AssertionError(f"conv_cache must be (2,{B},16,6,33), got {tuple(conv_cache.shape)}")
        if tuple(tra_cache.shape) != (2, 3, B, 8, 2):
            raise AssertionError(f"tra_cache must be (2,3,{B},8,2), got {tuple(tra_cache.shape)}")
        for g in range(2):
            for k, d in enumerate((1, 2, 4, 8)):
                if tuple(tcn_cache[g][k].shape) != (B, 16, 2 * d, 33):
                    raise AssertionError(f"tcn_cache[{g}][{k}] must be ({B},16,{2 * d},33)")
        eng = self.engine(spec.device)
        key = (spec.device.index, B)
        state = self._scratch.get(key)
        if state is None:
            state = eng.new_state(B)
            self._scratch[key] = state
        eng.stream_import(state, conv_cache, tra_cache, tcn_cache)
        out = eng.stream_step(state, spec)
        eng.stream_export(state, conv_cache, tra_cache, tcn_cache)
        return out, conv_cache, tra_cache, tcn_cache

    # ---- native streaming: state stays in the library's ring layout on the device ---------------
    def init_state(self, nstreams, device="cuda"):
        dev = torch.device(device)
        return self.engine(dev).new_state(nstreams)

    def step(self, spec_t, state):
        """spec_t (N,257,n,2), n >= 1 new frames per stream; state from init_state (updated in place)."""
        return self.engine(spec_t.device).stream_step(state, spec_t)
