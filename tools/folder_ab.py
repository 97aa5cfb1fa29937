import json, os, sys, shutil, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from scipy.io import wavfile
from gtcrn_micro_amd.infer import enhance_folder
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
root = tempfile.mkdtemp(prefix="gtcrn_folder_", dir="/dev/shm")
noisy, clean = os.path.join(root, "noisy"), os.path.join(root, "clean")
os.makedirs(noisy); os.makedirs(clean)
rng = np.random.default_rng(47)
lens = rng.integers(32000, 160001, 512)
for k, L in enumerate(lens):
    wavfile.write(os.path.join(noisy, f"mix_fileid_{k}.wav"), 16000, np.clip(rng.standard_normal(int(L)) * 3000, -32768, 32767).astype(np.int16))
    wavfile.write(os.path.join(clean, f"clean_fileid_{k}.wav"), 16000, np.zeros(int(L), np.int16))
ck = os.path.join(ROOT, "tests", "golden", "params_dns3.f32")
res = {}
for rep in range(3):
    for key, kw in (("serial", dict(pipeline=False)), ("pipe_io1", dict(io_threads=1)), ("pipe_io2", dict(io_threads=2)), ("pipe_io4", dict(io_threads=4)), ("pipe_io8", dict(io_threads=8))):
        st = {}
        enhance_folder(noisy, clean, os.path.join(root, "enh_" + key), ck, device=0, max_batch=64, stats=st, **kw)
        res.setdefault(key, []).append((round(st["wall_s"], 4), round(st.get("setup_s", 0.0), 4)))
print(json.dumps(res))
shutil.rmtree(root, ignore_errors=True)
