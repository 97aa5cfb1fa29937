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


def is_trainable(name):
    """The 248 tensors train.py hands to Adam: everything except BatchNorm running statistics and the
    frozen ERB filterbank (models/gtcrn_micro.py:27-33, requires_grad False)."""
    return not (name.endswith("running_mean") or name.endswith("running_var") or name.startswith("erb."))


class _RoundSTE(torch.autograd.Function):
    """x -> bf16(x) (round to nearest even) with a straight-through gradient: what a tensor STORED in bf16 and read
    back by its consumers looks like to the rest of the graph."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class TorchPort:
    def __init__(self, blob, train=False, dtype=torch.float32, store=None):
        """train=True: BatchNorm uses batch statistics and updates the running ones in place
        (nn.BatchNorm2d in .train() mode, momentum 0.1), and trainable tensors require grad.
        dtype=torch.float64 runs the same graph in double: the "truth" that fp32 results scatter around.
        store="bf16": the graph of the HIP trainer's bf16 storage mode (BASELINE configs[3]) -- every tensor the
        trainer writes out (features, conv outputs centred on the running mean, activations, block outputs, decoder
        sums, the mask) is rounded to bf16 where it is produced and its consumers read the rounded value; arithmetic in
        `dtype`.  With dtype=float64 this is the truth for THAT network: bf16-activation training computes the
        gradient of a slightly different function than the fp32 network (tests/reports/bf16_storage_ablation.py)."""
        self.p = {k: v.to(dtype) for k, v in blob_to_dict(blob).items()}
        self.dtype = dtype
        self.store = store
        self.train = bool(train)
        if self.train:
            for k, v in self.p.items():
                if is_trainable(k):
                    v.requires_grad_(True)

    # conv + BatchNorm + activation (ConvBlock.forward, models/gtcrn_micro.py:163-164)
    def _r(self, x):
        return _RoundSTE.apply(x) if self.store == "bf16" else x

    def _bn(self, x, pre):
        p = self.p
        if self.store == "bf16":
            # the conv output is stored centred on the channel's mean of the previous step (the running mean before the
            # first one, csrc/train.cpp unit_fwd); the batch statistics are taken from the stored values
            rm = p[pre + ".running_mean"].detach().clone().view(1, -1, 1, 1)
            x = _RoundSTE.apply(x - rm) + rm
        return F.batch_norm(x, p[pre + ".running_mean"], p[pre + ".running_var"], p[pre + ".weight"],
                            p[pre + ".bias"], self.train, 0.1, 1e-5)

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
        h = self._r(F.prelu(self._bn(h, pre + ".point_bn1"), p[pre + ".point_act.weight"]))
        if deconv:
            h = F.conv_transpose2d(h, p[pre + ".depth_conv.weight"], p[pre + ".depth_conv.bias"], padding=(0, 1))
        else:
            h = F.conv2d(F.pad(h, [0, 0, 2, 0]), p[pre + ".depth_conv.weight"], p[pre + ".depth_conv.bias"],
                         padding=(0, 1), groups=16)
        h = self._r(F.prelu(self._bn(h, pre + ".depth_bn"), p[pre + ".depth_act.weight"]))
        h = self._r(self._bn(conv(h, p[pre + ".point_conv2.weight"], p[pre + ".point_conv2.bias"]), pre + ".point_bn2"))
        h = self._tra(h, pre + ".tra")[:, :, :x2.shape[2]]
        return self._r(torch.stack([h, x2], dim=2).flatten(1, 2))      # out[2c] = h[c], out[2c+1] = x2[c]

    def _tcn(self, x, pre, d):
        # TCN.forward (models/gtcrn_micro.py:290-310)
        p = self.p
        y = self._r(F.prelu(self._bn(F.conv2d(x, p[pre + ".conv1.weight"], p[pre + ".conv1.bias"]), pre + ".bn1"),
                            p[pre + ".act1.weight"]))
        y = F.conv2d(F.pad(y, [0, 0, 2 * d, 0]), p[pre + ".conv2.weight"], p[pre + ".conv2.bias"],
                     dilation=(d, 1), groups=16)
        y = self._r(F.prelu(self._bn(y, pre + ".bn2"), p[pre + ".act2.weight"]))
        y = self._bn(F.conv2d(y, p[pre + ".conv3.weight"], p[pre + ".conv3.bias"]), pre + ".bn3")
        return self._r(F.prelu(y + x, p[pre + ".act3.weight"]))

    def forward(self, spec, taps=None):
        """GTCRNMicro.forward (models/gtcrn_micro.py:506-532): (B,257,T,2) -> (B,257,T,2)."""
        if self.train:
            return self._forward(spec, taps)
        with torch.inference_mode():
            return self._forward(spec, taps)

    def _forward(self, spec, taps=None):
        p = self.p

        def tap(name, v):
            if taps is not None:
                taps[name] = v.detach().clone()
        spec = torch.as_tensor(spec).to(self.dtype)
        re, im = spec[..., 0].permute(0, 2, 1), spec[..., 1].permute(0, 2, 1)
        feat = torch.stack([torch.sqrt(re * re + im * im + 1e-12), re, im], dim=1)
        feat = self._r(torch.cat([feat[..., :65], F.linear(feat[..., 65:], p["erb.erb_fc.weight"])], dim=-1))
        x = self._r(F.conv2d(feat, p["sfe.depth_conv.weight"], padding=(0, 1), groups=3))
        skips = []
        for i in range(2):
            pre = f"encoder.en_convs.{i}"
            x = F.conv2d(x, p[pre + ".conv.weight"], p[pre + ".conv.bias"], stride=(1, 2), padding=(0, 2))
            x = self._r(F.prelu(self._bn(x, pre + ".bn"), p[pre + ".act.weight"]))
            skips.append(x)
            tap(f"en{i}", x)
        for i in range(2, 5):
            x = self._gtconv(x, f"encoder.en_convs.{i}", False)
            skips.append(x)
            tap(f"en{i}", x)
        for g in (1, 2):
            for k in range(4):
                x = self._tcn(x, f"gtcn{g}.blocks.{k}", 1 << k)
            tap(f"gtcn{g}", x)
        for i in range(3):
            x = self._gtconv(self._r(x + skips[4 - i]), f"decoder.de_convs.{i}", True)
            tap(f"de{i}", x)
        pre = "decoder.de_convs.3"
        x = F.conv_transpose2d(self._r(x + skips[1]), p[pre + ".conv.weight"], p[pre + ".conv.bias"], stride=(1, 2), padding=(0, 2))
        x = self._r(F.prelu(self._bn(x, pre + ".bn"), p[pre + ".act.weight"]))
        tap("de3", x)
        pre = "decoder.de_convs.4"
        x = F.conv_transpose2d(self._r(x + skips[0]), p[pre + ".conv.weight"], p[pre + ".conv.bias"], stride=(1, 2), padding=(0, 2))
        m = self._r(torch.tanh(self._bn(x, pre + ".bn")))
        tap("de4", m)
        m = torch.cat([m[..., :65], F.linear(m[..., 65:], p["erb.ierb_fc.weight"])], dim=-1)   # (B,2,T,257)
        out_re = re * m[:, 0] - im * m[:, 1]
        out_im = im * m[:, 0] + re * m[:, 1]
        return torch.stack([out_re, out_im], dim=-1).permute(0, 2, 1, 3)

    # ---- train step (train.py:239-288): HybridLoss (loss.py:30-71) and the gradients ----------------
    @staticmethod
    def hybrid_loss(pred, true):
        pr, pi, tr, ti = pred[..., 0], pred[..., 1], true[..., 0], true[..., 1]
        pm = torch.sqrt(pr ** 2 + pi ** 2 + 1e-12)
        tm = torch.sqrt(tr ** 2 + ti ** 2 + 1e-12)
        mse = F.mse_loss
        ri = mse(pr / pm ** 0.7, tr / tm ** 0.7) + mse(pi / pm ** 0.7, ti / tm ** 0.7)
        mag = mse(pm ** 0.3, tm ** 0.3)
        win = torch.hann_window(512).pow(0.5).to(pred.device)
        yp = torch.istft(torch.complex(pr, pi), 512, 256, 512, window=win)
        yt = torch.istft(torch.complex(tr, ti), 512, 256, 512, window=win)
        yt = torch.sum(yt * yp, -1, keepdim=True) * yt / (torch.sum(yt ** 2, -1, keepdim=True) + 1e-8)
        sisnr = -torch.log10(torch.norm(yt, dim=-1, keepdim=True) ** 2 /
                             (torch.norm(yp - yt, dim=-1, keepdim=True) ** 2 + 1e-8) + 1e-8).mean()
        return 30 * ri + 70 * mag + sisnr

    def blob(self):
        man = json.load(open(_MANIFEST))
        return np.concatenate([self.p[n].detach().numpy().astype(np.float32).ravel() for n, _, _ in man["tensors"]])

    def grads_blob(self):
        """d loss / d parameter in the canonical blob layout (zeros in the slots of buffers)."""
        man = json.load(open(_MANIFEST))
        out = []
        for n, s, _ in man["tensors"]:
            g = self.p[n].grad
            out.append((g if g is not None else torch.zeros(s)).detach().numpy().astype(np.float32).ravel())
        return np.concatenate(out)

    def train_step(self, noisy_spec, clean_spec):
        """One forward/backward in train mode; returns (enh, loss, d loss/d enh, grads blob)."""
        assert self.train
        for v in self.p.values():
            v.grad = None
        enh = self.forward(torch.as_tensor(noisy_spec, dtype=torch.float32))
        enh.retain_grad()
        loss = self.hybrid_loss(enh, torch.as_tensor(clean_spec, dtype=torch.float32))
        loss.backward()
        return enh.detach().numpy(), float(loss), enh.grad.numpy(), self.grads_blob()

    def backward_from(self, spec, grad_enh):
        """Gradients for a given upstream gradient (the model's backward alone, without the loss)."""
        assert self.train
        for v in self.p.values():
            v.grad = None
        enh = self.forward(torch.as_tensor(spec).to(self.dtype))
        enh.backward(torch.as_tensor(grad_enh).to(self.dtype))
        return enh.detach().numpy(), self.grads_blob()

    @torch.inference_mode()
    def enhance(self, wave, window):
        """infer.py:60-76 for a batch of equal-length clips (the caller loop of the reference)."""
        wave = torch.as_tensor(wave, dtype=torch.float32)
        window = torch.as_tensor(window, dtype=torch.float32)
        spec = torch.view_as_real(torch.stft(wave, 512, 256, 512, window, return_complex=True))
        out = self.forward(spec)
        return torch.istft(torch.view_as_complex(out.contiguous()), 512, 256, 512, window)
