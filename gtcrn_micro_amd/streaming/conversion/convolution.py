"""Mirrors of the reference's streaming conv wrappers (``streaming/conversion/convolution.py``).

Same constructor arguments, state_dict keys (``Conv2d.weight`` / ``ConvTranspose2d.weight`` ...) and
``forward(x, cache) -> (output, out_cache)`` contract as ``StreamConv2d`` (:62-119) and
``StreamConvTranspose2d`` (:122-253); the arithmetic is the generic causal-conv HIP kernel behind
``gtcrn_stream_conv2d``.  ``StreamConvTranspose2d`` keeps a ``Conv2d`` parameter holder whose weight is the
permuted + flipped one that ``convert_to_stream`` produces (convert.py:35-48), exactly like the reference.
Frequency stride > 1 is not supported (the model never uses it; the reference's branch for it allocates on
the CPU, :222).  ``StreamConv1d`` (:10-59, unused by the model) runs through the same kernel with one
frequency bin.
"""
import torch
import torch.nn as nn

from ... import _lib


def _pair(v, what):
    if isinstance(v, int):
        return v, v
    if isinstance(v, (list, tuple)):
        return tuple(v)
    raise ValueError(f"Invalid {what}!")


class StreamConv1d(nn.Module):
    """``forward(x [bs,C,T], cache [bs,C,(K-1)*dilation]) -> (output, out_cache)``: conv over ``cat([cache, x])``
    and, like the reference (:52-59), ``out_cache = cat([cache, x])[..., 1:]``."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert padding == 0, "Padding must be 0 to keep it causal!"
        if stride != 1:
            raise NotImplementedError("stride 1 only")
        self.Conv1d = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                dilation=dilation, groups=groups, bias=bias)

    def forward(self, x, cache):
        c = self.Conv1d
        y, _ = _lib.stream_conv2d(x.unsqueeze(-1), cache.unsqueeze(-1), c.weight.unsqueeze(-1), c.bias,
                                  c.kernel_size[0], 1, c.dilation[0], 1, 0, c.groups, transposed=False)
        return y.squeeze(-1), torch.cat([cache, x], dim=-1)[..., 1:]


class StreamConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.T_pad, self.F_pad = _pair(padding, "padding size")
        assert self.T_pad == 0, "Time padding must be 0 to keep it causal!"
        if _pair(stride, "stride") != (1, 1):
            raise NotImplementedError("stride 1 only")
        self.Conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                dilation=dilation, groups=groups, bias=bias)

    def forward(self, x, cache):
        c = self.Conv2d
        return _lib.stream_conv2d(x, cache, c.weight, c.bias, c.kernel_size[0], c.kernel_size[1], c.dilation[0],
                                  c.dilation[1], self.F_pad, c.groups, transposed=False)


class StreamConvTranspose2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.T_size, self.F_size = _pair(kernel_size, "kernel size")
        self.T_stride, self.F_stride = _pair(stride, "Stride")
        assert self.T_stride == 1, f"Time stride must be 1 in deconv. Got {self.T_stride} instead"
        self.T_pad, self.F_pad = _pair(padding, "padding size")
        assert self.T_pad == 0, f"Padding must be 0 in deconv. Got {self.T_pad} instead"
        self.T_dilation, self.F_dilation = _pair(dilation, "dilation size")
        if self.F_stride != 1:
            raise NotImplementedError("frequency stride 1 only")
        # like the reference: a Conv2d holds the permuted + flipped weight set by convert_to_stream()
        self.ConvTranspose2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride=(1, 1), padding=(0, 0),
                                         dilation=dilation, groups=groups, bias=bias)

    def forward(self, x, cache):
        c = self.ConvTranspose2d
        # Conv2d over the frequency-padded input with the flipped weights (convolution.py:243-252):
        # pad = (kF-1)*dF - F_pad on both sides
        pad = (self.F_size - 1) * self.F_dilation - self.F_pad
        return _lib.stream_conv2d(x, cache, c.weight, c.bias, self.T_size, self.F_size, self.T_dilation,
                                  self.F_dilation, pad, c.groups, transposed=False)
