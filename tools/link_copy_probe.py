#!/usr/bin/env python3
"""Both directions of the host link at once: DMA engines (tensor.copy_) against the copy kernel (gtcrn_link_copy) for
either direction, at the served pipeline's transfer size (65.5 MB each way).

    python tools/link_copy_probe.py [--mb 65.5] [--wgs 16,32,64,128]

Prints GB/s per direction for: each direction alone (DMA / kernel), and the four duplex combinations."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=float, default=65.536)
    ap.add_argument("--wgs", default="16,32,64,128")
    a = ap.parse_args()
    import torch
    import gtcrn_micro_amd as G
    n = int(a.mb * 1e6 / 4) // 4 * 4
    hin = torch.randn(n).pin_memory()
    hout = torch.empty(n).pin_memory()
    din = torch.empty(n, device="cuda")
    dout = torch.randn(n, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    nbytes = n * 4

    def timeit(fn, reps=20, warm=30):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return nbytes * reps / (time.perf_counter() - t0) / 1e9

    def h2d_dma():
        with torch.cuda.stream(s1):
            din.copy_(hin, non_blocking=True)

    def d2h_dma():
        with torch.cuda.stream(s2):
            hout.copy_(dout, non_blocking=True)

    print(f"{nbytes / 1e6:.1f} MB per direction")
    print(f"alone   H2D dma {timeit(h2d_dma):6.1f} GB/s   D2H dma {timeit(d2h_dma):6.1f} GB/s")
    for w in [int(x) for x in a.wgs.split(",")]:
        def h2d_k():
            G.link_copy(din, hin, workgroups=w, stream=s1)

        def d2h_k():
            G.link_copy(hout, dout, workgroups=w, stream=s2)
        print(f"wgs {w:4d}: alone H2D kernel {timeit(h2d_k):6.1f}  D2H kernel {timeit(d2h_k):6.1f} GB/s (each direction)")
        for name, f1, f2 in (("dma + dma      ", h2d_dma, d2h_dma), ("dma  + D2H kern", h2d_dma, d2h_k),
                             ("H2D kern + dma ", h2d_k, d2h_dma), ("kern + kern    ", h2d_k, d2h_k)):
            def both():
                f1(); f2()
            print(f"   duplex {name}: {timeit(both):6.1f} GB/s each way")
    torch.cuda.synchronize()
    assert torch.equal(hout, dout.cpu()) and torch.equal(din.cpu(), hin)
    print("copies verified")


if __name__ == "__main__":
    main()
