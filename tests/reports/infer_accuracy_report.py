#!/usr/bin/env python3
"""Diagnostic (GPU box): accuracy of the eval-mode forward against the SAME graph evaluated in float64
(oracle/torch_port.py, dtype=float64), next to the reference's own fp32 result stored in the fixtures.

    python tests/reports/infer_accuracy_report.py
"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gtcrn_micro_amd import Engine
from oracle.torch_port import TorchPort
torch.set_num_threads(min(8, torch.get_num_threads()))
for tag in ("dns3", "rand"):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"offline_{tag}_T17.npz"))
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", f"params_{tag}.f32"), np.float32)
    spec = g["spec"][None] if g["spec"].ndim == 3 else g["spec"]
    t64 = TorchPort(blob, dtype=torch.float64).forward(torch.from_numpy(spec)).numpy()
    ours = Engine(blob, 0).forward_spec(torch.from_numpy(spec).cuda()).cpu().numpy()
    ref = g["spec_enh"][None] if g["spec_enh"].ndim == 3 else g["spec_enh"]
    sc = np.abs(t64).max()
    print(tag, "HIP vs fp64: %.2e   reference fp32 vs fp64: %.2e" % (np.abs(ours - t64).max() / sc, np.abs(ref - t64).max() / sc))
    # a larger random case (4 utterances x 64 frames): max and rms error of the HIP forward and of the fp32 PyTorch-CPU
    # port of the same graph, both against float64
    rng = np.random.default_rng(5)
    spec2 = (rng.standard_normal((4, 257, 64, 2)) * 0.3).astype(np.float32)
    t64 = TorchPort(blob, dtype=torch.float64).forward(torch.from_numpy(spec2)).numpy()
    t32 = TorchPort(blob).forward(torch.from_numpy(spec2)).numpy()
    ours = Engine(blob, 0).forward_spec(torch.from_numpy(spec2).cuda()).cpu().numpy()
    sc = np.abs(t64).max()
    print(tag, "B=4,T=64  HIP vs fp64: max %.2e rms %.2e   PyTorch-CPU fp32 vs fp64: max %.2e rms %.2e   [lib variant: %s]" % (
        np.abs(ours - t64).max() / sc, np.sqrt(np.mean((ours - t64) ** 2)) / sc,
        np.abs(t32 - t64).max() / sc, np.sqrt(np.mean((t32 - t64) ** 2)) / sc, os.environ.get("GTCRN_LIB_VARIANT", "default")))
