"""Checker of the int8-weight / fp16-activation variant (BASELINE configs[4]) -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference ships no quantised model, calibration data or TensorFlow, so there is
nothing of the reference's to compare numbers with.  This file restates, as plain PyTorch-CPU ops, the contract
written in include/gtcrn_micro_hip.h (gtcrn_forward_spec_quant) from the reference's own pieces:

  * graph: GTCRNMicro.forward in eval mode (models/gtcrn_micro.py:506-532), BatchNorm folded into the preceding
    conv (what the exported graph holds), PReLU as the ReLU composite of `-rtpo PReLU` (same function);
  * weights: symmetric int8 per OUTPUT channel (onnx2tf `-oiqt -qt per-channel`, scripts/onnx2tf.sh:50-64),
    scale = max|w| / 127, consumed as fp16(q * scale);
  * activations: rounded to fp16 (RNE) where a layer produces them; sums and products in fp32;
  * optional int8 boundary: x_q = clip(round(x / (scale/255)), -128, 127) (tflite_infer.py:79-92, zero point 0;
    scale = 19.944473, streaming/tflite/calib_scale.txt; utils/calibration_data.py:97-106).

The HIP kernels accumulate in a different order than ATen, so values differ by fp32 rounding BEFORE each fp16
rounding: a fraction of the elements lands on the neighbouring fp16 value (2^-11 relative).  Tests state the tolerance.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .torch_port import blob_to_dict

CALIB_SCALE = 19.944473266601562          # streaming/tflite/calib_scale.txt:1


def h16(x):
    return x.half().float()


def int8_boundary(x, scale):
    step = np.float32(scale / 255.0)
    return torch.clamp(torch.round(x / step), -128, 127) * step


def quant_per_out_channel(w, out_dim=0):
    """w -> fp16(int8 * scale), one scale per index of `out_dim`."""
    w = w.double()
    dims = [d for d in range(w.dim()) if d != out_dim]
    mx = w.abs().amax(dim=dims, keepdim=True)
    scale = (mx / 127.0).float().double()
    q = torch.where(scale > 0, torch.clamp(torch.round(w / torch.where(scale > 0, scale, torch.ones_like(scale))),
                                           -127, 127), torch.zeros_like(w))
    return h16((q * scale).float())


class QuantPort:
    def __init__(self, blob):
        p = {k: v.double() for k, v in blob_to_dict(blob).items()}
        self.p = p
        self.w, self.b = {}, {}

        def fold(conv, bn, transposed):
            s = p[bn + ".weight"] / torch.sqrt(p[bn + ".running_var"] + 1e-5)
            w = p[conv + ".weight"]
            w = w * (s.view(1, -1, 1, 1) if transposed else s.view(-1, 1, 1, 1))
            bias = p.get(conv + ".bias")
            bias = torch.zeros_like(s) if bias is None else bias
            self.w[conv] = quant_per_out_channel(w.float(), 1 if transposed else 0)
            self.b[conv] = ((bias - p[bn + ".running_mean"]) * s + p[bn + ".bias"]).float()
        for i in range(2):
            fold(f"encoder.en_convs.{i}.conv", f"encoder.en_convs.{i}.bn", False)
        for pre, dec in [(f"encoder.en_convs.{i}", False) for i in (2, 3, 4)] + \
                        [(f"decoder.de_convs.{i}", True) for i in range(3)]:
            fold(pre + ".point_conv1", pre + ".point_bn1", dec)
            fold(pre + ".depth_conv", pre + ".depth_bn", dec)
            fold(pre + ".point_conv2", pre + ".point_bn2", dec)
            self.w[pre + ".tra.depth_conv"] = quant_per_out_channel(p[pre + ".tra.depth_conv.weight"].float())
            self.w[pre + ".tra.point_conv"] = quant_per_out_channel(p[pre + ".tra.point_conv.weight"].float())
        for g in (1, 2):
            for k in range(4):
                pre = f"gtcn{g}.blocks.{k}"
                for j in (1, 2, 3):
                    fold(f"{pre}.conv{j}", f"{pre}.bn{j}", False)
        fold("decoder.de_convs.3.conv", "decoder.de_convs.3.bn", True)
        fold("decoder.de_convs.4.conv", "decoder.de_convs.4.bn", True)
        self.w["erb"] = quant_per_out_channel(p["erb.erb_fc.weight"].float())
        self.w["ierb"] = quant_per_out_channel(p["erb.ierb_fc.weight"].float())
        self.w["sfe"] = quant_per_out_channel(p["sfe.depth_conv.weight"].float())
        self.f = {k: v.float() for k, v in p.items()}

    def _tra(self, v, pre):
        f = self.f
        e = (v * v).mean(dim=3)
        y = F.conv1d(F.pad(e, [2, 0]), self.w[pre + ".depth_conv"], f[pre + ".depth_conv.bias"], groups=8)
        g = h16(torch.sigmoid(F.conv1d(y, self.w[pre + ".point_conv"], f[pre + ".point_conv.bias"])))
        return h16(v * g.unsqueeze(-1))

    def _gtconv(self, x, pre, deconv):
        f, w, b = self.f, self.w, self.b
        x1, x2 = x[:, :8], x[:, 8:]
        conv = F.conv_transpose2d if deconv else F.conv2d
        h = h16(F.prelu(conv(x1, w[pre + ".point_conv1"], b[pre + ".point_conv1"]), f[pre + ".point_act.weight"]))
        if deconv:
            h = F.conv_transpose2d(h, w[pre + ".depth_conv"], b[pre + ".depth_conv"], padding=(0, 1))[:, :, :x.shape[2]]
        else:
            h = F.conv2d(F.pad(h, [0, 0, 2, 0]), w[pre + ".depth_conv"], b[pre + ".depth_conv"], padding=(0, 1), groups=16)
        h = h16(F.prelu(h, f[pre + ".depth_act.weight"]))
        v = h16(conv(h, w[pre + ".point_conv2"], b[pre + ".point_conv2"]))
        v = self._tra(v, pre + ".tra")
        return torch.stack([v, x2], dim=2).flatten(1, 2)

    def _tcn(self, x, pre, d):
        f, w, b = self.f, self.w, self.b
        y = h16(F.prelu(F.conv2d(x, w[pre + ".conv1"], b[pre + ".conv1"]), f[pre + ".act1.weight"]))
        y = F.conv2d(F.pad(y, [0, 0, 2 * d, 0]), w[pre + ".conv2"], b[pre + ".conv2"], dilation=(d, 1), groups=16)
        y = h16(F.prelu(y, f[pre + ".act2.weight"]))
        y = F.conv2d(y, w[pre + ".conv3"], b[pre + ".conv3"])
        return h16(F.prelu(y + x, f[pre + ".act3.weight"]))

    @torch.inference_mode()
    def forward(self, spec, in_scale=0.0, out_scale=0.0):
        f, w, b = self.f, self.w, self.b
        spec = torch.as_tensor(spec, dtype=torch.float32)
        if in_scale > 0:
            spec = int8_boundary(spec, in_scale)
        spec = h16(spec)
        re, im = spec[..., 0].permute(0, 2, 1), spec[..., 1].permute(0, 2, 1)
        feat = torch.stack([h16(torch.sqrt(re * re + im * im + 1e-12)), re, im], dim=1)
        feat = torch.cat([feat[..., :65], h16(F.linear(feat[..., 65:], w["erb"]))], dim=-1)
        x = h16(F.conv2d(feat, w["sfe"], padding=(0, 1), groups=3))
        skips = []
        for i in range(2):
            pre = f"encoder.en_convs.{i}"
            x = F.conv2d(x, w[pre + ".conv"], b[pre + ".conv"], stride=(1, 2), padding=(0, 2))
            x = h16(F.prelu(x, f[pre + ".act.weight"]))
            skips.append(x)
        for i in range(2, 5):
            x = self._gtconv(x, f"encoder.en_convs.{i}", False)
            skips.append(x)
        for g in (1, 2):
            for k in range(4):
                x = self._tcn(x, f"gtcn{g}.blocks.{k}", 1 << k)
        for i in range(3):
            x = self._gtconv(h16(x + skips[4 - i]), f"decoder.de_convs.{i}", True)
        pre = "decoder.de_convs.3"
        x = F.conv_transpose2d(h16(x + skips[1]), w[pre + ".conv"], b[pre + ".conv"], stride=(1, 2), padding=(0, 2))
        x = h16(F.prelu(x, f[pre + ".act.weight"]))
        pre = "decoder.de_convs.4"
        m = F.conv_transpose2d(h16(x + skips[0]), w[pre + ".conv"], b[pre + ".conv"], stride=(1, 2), padding=(0, 2))
        m = h16(torch.tanh(m))
        m = torch.cat([m[..., :65], h16(F.linear(m[..., 65:], w["ierb"]))], dim=-1)
        out = torch.stack([h16(re * m[:, 0] - im * m[:, 1]), h16(im * m[:, 0] + re * m[:, 1])], dim=-1).permute(0, 2, 1, 3)
        if out_scale > 0:
            out = int8_boundary(out, out_scale)
        return out.numpy()
