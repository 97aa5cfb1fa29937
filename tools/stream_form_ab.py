import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gtcrn_micro_amd import Engine
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
eng = Engine(np.fromfile(os.path.join(ROOT, "tests", "golden", "params_dns3.f32"), dtype=np.float32), 0)
# forms to alternate (gtcrn_stream_form): default 0 (one launch, form by stream count) against 1 (three launches);
# `stream_form_ab.py 2 3` = k_stream_ms (four streams per workgroup) against k_stream_wide (seven)
FORMS = [int(a) for a in sys.argv[1:3]] if len(sys.argv) >= 3 else [0, 1]
NS = [int(a) for a in sys.argv[3:]] or [1024, 4096, 16384, 65536]
res = {}
for N in NS:
    spec = (torch.randn(N, 8, 257, 2, device="cuda") * 0.3).permute(0, 2, 1, 3)
    out = torch.empty((N, 1, 257, 2), device="cuda").permute(0, 2, 1, 3)
    st = eng.new_state(N)
    eng.reserve(N, 1)
    calls = 100 if N <= 4096 else 40
    r = {}
    for form in FORMS + FORMS:
        eng.stream_form(form)
        for t in range(3):
            eng.stream_step(st, spec[:, :, t:t + 1], out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(calls):
            eng.stream_step(st, spec[:, :, t % 8:t % 8 + 1], out=out)
        torch.cuda.synchronize()
        r.setdefault(f"form{form}_ms", []).append(round((time.perf_counter() - t0) / calls * 1e3, 4))
    eng.stream_form(0)
    res[N] = r
    del spec, out, st
    torch.cuda.empty_cache()
print(json.dumps(res))
