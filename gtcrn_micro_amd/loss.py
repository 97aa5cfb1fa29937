"""Mirror of the reference's ``gtcrn_micro.loss.HybridLoss`` (loss.py:8-71) for tensors on the GPU.

Compressed real/imaginary/magnitude MSE (30/30/70) plus the SI-SNR of the sqrt-Hann iSTFTs, value
and gradient from the fused HIP kernels (gtcrn_train_loss); no torch-op fallback.  Same constructor arguments as the reference (accepted, and
like there the transform sizes are fixed at 512/256/512)."""
import torch
import torch.nn as nn

from . import _lib

_trainers = {}


def _trainer(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _trainers:
        _trainers[idx] = _lib.Trainer(idx)
    return _trainers[idx]


class _FusedHybridLoss(torch.autograd.Function):
    """Value and gradient from the HIP kernels (gtcrn_train_loss): one pass for the three spectral terms, the two
    sqrt-Hann iSTFTs, three sums per utterance for the SI-SNR, and the iSTFT adjoint for its gradient."""

    @staticmethod
    def forward(ctx, pred, true):
        loss, grad = _trainer(pred.device).hybrid_loss(pred.detach(), true.detach(), want_grad=pred.requires_grad)
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, go):
        return (ctx.grad * go if ctx.grad is not None else None), None


class HybridLoss(nn.Module):
    def __init__(self, n_fft=512, hop_len=256, win_len=512, compress_factor=0.3, eps=1e-12, lamda_ri=30,
                 lamda_mag=70):
        super().__init__()
        self.n_fft, self.hop_len, self.win_len = n_fft, hop_len, win_len
        self.c, self.eps, self.lamda_ri, self.lamda_mag = compress_factor, eps, lamda_ri, lamda_mag
        self.register_buffer("window", torch.hann_window(512).pow(0.5), persistent=False)

    def forward(self, pred_stft, true_stft):
        """Always the fused HIP kernels (gtcrn_train_loss).  Inputs they do not take raise -- there is no torch-op
        fallback in the product (the same loss written as torch ops lives in oracle/torch_port.py, the checker)."""
        if not pred_stft.is_cuda or not true_stft.is_cuda:
            raise _lib.GtcrnError("HybridLoss needs CUDA (ROCm) tensors: this implementation has no CPU path")
        if pred_stft.dtype != torch.float32 or true_stft.dtype != torch.float32:
            raise _lib.GtcrnError("HybridLoss takes float32 spectrograms")
        if true_stft.requires_grad:
            raise _lib.GtcrnError("HybridLoss: a gradient w.r.t. the target is not supported (the reference's train "
                                  "step never asks for one, train.py:267): detach the clean spectrogram")
        if pred_stft.dim() != 4 or pred_stft.shape[2] < 2:
            raise _lib.GtcrnError("HybridLoss needs (B,257,T,2) spectrograms with T >= 2 frames (torch.istft's limit)")
        B = pred_stft.shape[0]
        if B <= 1024:
            return _FusedHybridLoss.apply(pred_stft, true_stft)
        # more utterances than one launch takes: the loss is a batch mean, so equal-weight chunks combine exactly
        parts, n = [], 0
        for lo in range(0, B, 1024):
            p, t = pred_stft[lo:lo + 1024], true_stft[lo:lo + 1024]
            parts.append(_FusedHybridLoss.apply(p, t) * (p.shape[0] / B))
        return torch.stack(parts).sum()
