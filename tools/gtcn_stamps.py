"""Per-TCN-block cycles of the GTCN phase of k_stream_wide (diagnostic build with -DGT_STAMPS -DGT_STAMPS_GTCN):
    GT_STAMPS_FLAGS="-DGT_STAMPS_GTCN [-DGT_EXP_SAMESTATE]" python tools/gtcn_stamps.py [N]"""
import os, sys
os.environ["GTCRN_LIB_VARIANT"] = "stamps"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gtcrn_micro_amd.build import build_native
build_native(stamps=True, force=True)
from gtcrn_micro_amd import Engine
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
eng = Engine(np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32), 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 7168
spec = (torch.randn(N, 40, 257, 2, device="cuda") * 0.3).permute(0, 2, 1, 3)
st = eng.new_state(N)
eng.stream_form(3)
for t in range(20):
    eng.stream_step(st, spec[:, :, t:t + 1])
eng.debug_enable(2)
for t in range(20, 40):
    eng.stream_step(st, spec[:, :, t:t + 1])
torch.cuda.synchronize()
stp = eng.stamps(0, N).astype(np.float64)[: (N + 6) // 7]
avg = stp.mean(axis=0)
print(os.environ.get("GT_STAMPS_FLAGS", ""), "TCN blocks, stack 1:", [int(x) for x in avg[:4]], "stack 2:", [int(x) for x in avg[4:8]], "sum", int(avg[:8].sum()))
