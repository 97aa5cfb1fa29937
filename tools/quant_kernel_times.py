#!/usr/bin/env python3
"""Per-kernel times of the int8-weight / fp16-activation variant next to the fp32 path (B = 256 x 4 s, wave -> wave)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from gtcrn_micro_amd import Engine
    params = np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32)
    eng = Engine(params, 0)
    torch.manual_seed(43)
    wave = torch.randn(256, 64000, device="cuda") * 0.1
    win = torch.hann_window(512).pow(0.5).cuda()
    for name, fn in (("fp32", lambda: eng.forward_wave(wave, win)), ("quant", lambda: eng.forward_wave_quant(wave, win))):
        for _ in range(5):
            fn()
        eng.timing_enable(True)
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        k = eng.timing_read()
        eng.timing_enable(False)
        tot = sum(v[0] for v in k.values())
        print(name, " ".join(f"{n}={v[0] * 1e3:.1f}us" for n, v in k.items()), f"sum={tot * 1e3:.1f}us")


if __name__ == "__main__":
    main()
