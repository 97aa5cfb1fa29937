"""Functional PyTorch-CPU port of the GTCRN-Micro hot path (test infrastructure + CPU baseline).

TEST INFRASTRUCTURE ONLY, like everything under oracle/: never imported by the product.
Why it exists next to the C oracle: the reference executes on PyTorch ATen (mkldnn conv,
native BN/PReLU, pocketfft), and its Python cannot travel to the GPU box.  This port issues the
same ATen op sequence from the flat parameter blob, so that bench.py's `cpu_baseline` times the
reference's own CPU arithmetic (kind "port") rather than a slower scalar restatement.
Parity status: PINNED against tests/golden (tests/test_oracle_golden.py).

Written from the layer list in SURVEY.md section 2a / Appendix A; cites reference lines.
"""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_MANIFEST = os.path.join(_HERE, "..", "tests", "golden", "params_manifest.json")


def blob_to_dict(blob):
    man = json.load(open(_MANIFEST))
    blob = np.asarray(blob, np.float32)
    return {n: torch.from_numpy(blob[o:o + int(np.prod(s))].reshape(s).copy()) for n, s, o in man["tensors"]}


class TorchPort:
    def __init__(self, blob):
        self.p = blob_to_dict(blob)

    # conv + eval BatchNorm + activation (ConvBlock.forward, models/gtcrn_micro.py:163-164)
    def _bn(self, x, pre):
        p = self.p
        return F.batch_norm(x, p[pre + ".running_mean"], p[pre + ".running_var"], p[pre + ".weight"],
                            p[pre + ".bias"], False, 0.1, 1e-5)

    def _tra(self, x, pre):
        # TRALite.forward (models/gtcrn_micro.py:122-139), zero cache
        p = self.p
        e = (x * x).mean(dim=3)
        y = F.conv1d(F.pad(e, [2, 0]), p[pre + ".depth_conv.weight"], p[pre + ".depth_conv.bias"], groups=8)
        g = torch.sigmoid(F.conv1d(y, p[pre + ".point_conv.weight"], p[pre + ".point_conv.bias"]))
        return x * g.unsqueeze(-1)

    def _gtconv(self, x, pre, deconv):
        # GTConvBlock.forward (models/gtcrn_micro.py:229-253)
        p = self.p
        x1, x2 = x[:, :8], x[:, 8:]
        conv = F.conv_transpose2d if deconv else F.conv2d
        h = conv(x1, p[pre + ".point_conv1.weight"], p[pre + ".point_conv1.bias"])
        h = F.prelu(self._bn(h, pre + ".point_bn1"), p[pre + ".point_act.weight"])
        if deconv:
            h = F.conv_transpose2d(h, p[pre + ".depth_conv.weight"], p[pre + ".depth_conv.bias"], padding=(0, 1))
        else:
            h = F.conv2d(F.pad(h, [0, 0, 2, 0]), p[pre + ".depth_conv.weight"], p[pre + ".depth_conv.bias"],
                         padding=(0, 1), groups=16)
        h = F.prelu(self._bn(h, pre + ".depth_bn"), p[pre + ".depth_act.weight"])
        h = self._bn(conv(h, p[pre + ".point_conv2.weight"], p[pre + ".point_conv2.bias"]), pre + ".point_bn2")
        h = self._tra(h, pre + ".tra")[:, :, :x2.shape[2]]
        return torch.stack([h, x2], dim=2).flatten(1, 2)      # out[2c] = h[c], out[2c+1] = x2[c]

    def _tcn(self, x, pre, d):
        # TCN.forward (models/gtcrn_micro.py:290-310)
        p = self.p
        y = F.prelu(self._bn(F.conv2d(x, p[pre + ".conv1.weight"], p[pre + ".conv1.bias"]), pre + ".bn1"),
                    p[pre + ".act1.weight"])
        y = F.conv2d(F.pad(y, [0, 0, 2 * d, 0]), p[pre + ".conv2.weight"], p[pre + ".conv2.bias"],
                     dilation=(d, 1), groups=16)
        y = F.prelu(self._bn(y, pre + ".bn2"), p[pre + ".act2.weight"])
        y = self._bn(F.conv2d(y, p[pre + ".conv3.weight"], p[pre + ".conv3.bias"]), pre + ".bn3")
        return F.prelu(y + x, p[pre + ".act3.weight"])

    @torch.inference_mode()
    def forward(self, spec):
        """GTCRNMicro.forward (models/gtcrn_micro.py:506-532): (B,257,T,2) -> (B,257,T,2)."""
        p = self.p
        spec = torch.as_tensor(spec, dtype=torch.float32)
        re, im = spec[..., 0].permute(0, 2, 1), spec[..., 1].permute(0, 2, 1)
        feat = torch.stack([torch.sqrt(re * re + im * im + 1e-12), re, im], dim=1)
        feat = torch.cat([feat[..., :65], F.linear(feat[..., 65:], p["erb.erb_fc.weight"])], dim=-1)
        x = F.conv2d(feat, p["sfe.depth_conv.weight"], padding=(0, 1), groups=3)
        skips = []
        for i in range(2):
            pre = f"encoder.en_convs.{i}"
            x = F.conv2d(x, p[pre + ".conv.weight"], p[pre + ".conv.bias"], stride=(1, 2), padding=(0, 2))
            x = F.prelu(self._bn(x, pre + ".bn"), p[pre + ".act.weight"])
            skips.append(x)
        for i in range(2, 5):
            x = self._gtconv(x, f"encoder.en_convs.{i}", False)
            skips.append(x)
        for g in (1, 2):
            for k in range(4):
                x = self._tcn(x, f"gtcn{g}.blocks.{k}", 1 << k)
        for i in range(3):
            x = self._gtconv(x + skips[4 - i], f"decoder.de_convs.{i}", True)
        pre = "decoder.de_convs.3"
        x = F.conv_transpose2d(x + skips[1], p[pre + ".conv.weight"], p[pre + ".conv.bias"], stride=(1, 2), padding=(0, 2))
        x = F.prelu(self._bn(x, pre + ".bn"), p[pre + ".act.weight"])
        pre = "decoder.de_convs.4"
        x = F.conv_transpose2d(x + skips[0], p[pre + ".conv.weight"], p[pre + ".conv.bias"], stride=(1, 2), padding=(0, 2))
        m = torch.tanh(self._bn(x, pre + ".bn"))
        m = torch.cat([m[..., :65], F.linear(m[..., 65:], p["erb.ierb_fc.weight"])], dim=-1)   # (B,2,T,257)
        out_re = re * m[:, 0] - im * m[:, 1]
        out_im = im * m[:, 0] + re * m[:, 1]
        return torch.stack([out_re, out_im], dim=-1).permute(0, 2, 1, 3)

    @torch.inference_mode()
    def enhance(self, wave, window):
        """infer.py:60-76 for a batch of equal-length clips (the caller loop of the reference)."""
        wave = torch.as_tensor(wave, dtype=torch.float32)
        window = torch.as_tensor(window, dtype=torch.float32)
        spec = torch.view_as_real(torch.stft(wave, 512, 256, 512, window, return_complex=True))
        out = self.forward(spec)
        return torch.istft(torch.view_as_complex(out.contiguous()), 512, 256, 512, window)
