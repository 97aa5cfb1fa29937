"""Mirror of the reference's ``gtcrn_micro.loss.HybridLoss`` (loss.py:8-71) for tensors on the GPU.

SURVEY.md section 8f row 2 keeps the loss in PyTorch-ROCm (it is the outer training loop's
business, not the model hot path): compressed real/imaginary/magnitude MSE (30/30/70) plus the
SI-SNR of the sqrt-Hann iSTFTs.  Same constructor arguments as the reference (accepted, and
like there the transform sizes are fixed at 512/256/512)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

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
        if not pred_stft.is_cuda:
            raise _lib.GtcrnError("HybridLoss needs CUDA (ROCm) tensors: this implementation has no CPU path")
        if pred_stft.dtype == torch.float32 and not true_stft.requires_grad \
                and 2 <= pred_stft.shape[2] and pred_stft.shape[0] <= 1024:
            return _FusedHybridLoss.apply(pred_stft, true_stft)
        # shapes the fused kernels do not take (more than 1024 utterances, a gradient w.r.t. the target): the same
        # loss as torch ops on the GPU
        return self.forward_torch(pred_stft, true_stft)

    def forward_torch(self, pred_stft, true_stft):
        """The same loss as a chain of torch ops (the reference's own formulation): the checker of the fused path,
        and the path for inputs the fused kernels do not take (CPU tensors, gradients w.r.t. the target)."""
        pr, pi = pred_stft[..., 0], pred_stft[..., 1]
        tr, ti = true_stft[..., 0], true_stft[..., 1]
        pm = torch.sqrt(pr ** 2 + pi ** 2 + 1e-12)
        tm = torch.sqrt(tr ** 2 + ti ** 2 + 1e-12)
        real_loss = F.mse_loss(pr / pm ** 0.7, tr / tm ** 0.7)
        imag_loss = F.mse_loss(pi / pm ** 0.7, ti / tm ** 0.7)
        mag_loss = F.mse_loss(pm ** 0.3, tm ** 0.3)
        win = self.window.to(pred_stft.device)
        y_pred = torch.istft(torch.complex(pr, pi), 512, 256, 512, window=win)
        y_true = torch.istft(torch.complex(tr, ti), 512, 256, 512, window=win)
        y_true = torch.sum(y_true * y_pred, dim=-1, keepdim=True) * y_true / (
            torch.sum(torch.square(y_true), dim=-1, keepdim=True) + 1e-8)
        sisnr = -torch.log10(torch.norm(y_true, dim=-1, keepdim=True) ** 2 /
                             (torch.norm(y_pred - y_true, dim=-1, keepdim=True) ** 2 + 1e-8) + 1e-8).mean()
        return 30 * (real_loss + imag_loss) + 70 * mag_loss + sisnr
