"""Bulk offline enhancement on MI355X: the counterpart of the reference's ``gtcrn_micro/infer.py``.

Same contract as ``infer.py:26-119``: every ``*.wav`` of the noisy folder (sorted) is enhanced with the
checkpoint, length-matched to its clean reference (zero-pad / crop, ``:98-102``), written as
``<uid>_enh.wav`` and listed in ``inf.scp`` / ``ref.scp`` (``:113-119``).  Differences, all on the host side:

* clips of equal length are batched through one fused wave->wave call (STFT -> model -> iSTFT on the GPU);
* several GPUs shard the sorted file list (one process per GPU, no collective; ``sharding.shard_range``);
* WAV I/O uses ``scipy.io.wavfile`` (16-bit PCM out, what libsndfile writes for the reference); a file that
  is not 16 kHz raises instead of being resampled (the reference needs librosa for that, ``:55-57``);
* configuration comes from command-line flags (the reference reads two OmegaConf YAML files, ``:27-28``).

    python -m gtcrn_micro_amd.infer --noisy-dir N --clean-dir C --enh-dir E --checkpoint best_model_dns3.tar
"""
import argparse
import os

import numpy as np


def extract_fileid(path):
    """DNS3 naming: ``..._fileid_<N>.wav`` <-> ``clean_fileid_<N>.wav`` (infer.py:17-22)."""
    base = os.path.basename(path)
    if "fileid_" not in base:
        return None
    return base.split("fileid_")[-1].split(".")[0]


def read_wav_f32(path):
    from scipy.io import wavfile
    fs, x = wavfile.read(path)
    if x.ndim > 1:
        x = x[:, 0]
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    elif x.dtype == np.int32:
        x = x.astype(np.float32) / 2147483648.0
    else:
        x = x.astype(np.float32)
    return fs, x


def write_wav_pcm16(path, x, fs=16000):
    from scipy.io import wavfile
    wavfile.write(path, fs, np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16))


def load_params(checkpoint):
    """``.tar``/``.pt`` checkpoint of the reference (``ckpt["model"]``, train.py:200-216) or a raw fp32 blob."""
    if checkpoint.endswith((".f32", ".bin")):
        return np.fromfile(checkpoint, dtype=np.float32)
    import torch
    from .models.gtcrn_micro import state_dict_to_blob
    ck = torch.load(checkpoint, map_location="cpu", weights_only=False)
    return state_dict_to_blob(ck["model"] if "model" in ck else ck)


def enhance_folder(noisy_dir, clean_dir, enh_dir, checkpoint, device=0, max_batch=64, rank=0, world=1):
    import torch
    from . import Engine
    from .sharding import shard_range

    os.makedirs(enh_dir, exist_ok=True)
    eng = Engine(load_params(checkpoint), device)
    win = torch.hann_window(512).pow(0.5).to(f"cuda:{device}")      # infer.py:65
    names = sorted(f for f in os.listdir(noisy_dir) if f.endswith("wav"))
    lo, hi = shard_range(len(names), world, rank)
    items = []
    for wav_name in names[lo:hi]:
        path = os.path.join(noisy_dir, wav_name)
        fs, x = read_wav_f32(path)
        if fs != 16000:
            raise AssertionError(f"{path}: sample rate {fs} != 16000 (resampling is not part of this path)")
        fileid = extract_fileid(path)
        if fileid is None:
            raise RuntimeError(f"Unable to extract: {path}")
        ref_path = os.path.join(clean_dir, f"clean_fileid_{fileid}.wav")
        if not os.path.exists(ref_path):
            raise FileNotFoundError(f"Clean file not found for clean_fileid_{fileid}.wav, fileid={fileid}:\n {ref_path}")
        fs_c, clean = read_wav_f32(ref_path)
        if fs_c != fs:
            raise AssertionError(f"{ref_path}: sample rate {fs_c} != {fs}")
        items.append((wav_name, x, ref_path, len(clean)))
    # batch clips of equal length through one fused call
    by_len = {}
    for i, it in enumerate(items):
        by_len.setdefault(len(it[1]), []).append(i)
    enhanced = [None] * len(items)
    for L, idxs in by_len.items():
        for k in range(0, len(idxs), max_batch):
            sel = idxs[k:k + max_batch]
            wave = torch.from_numpy(np.stack([items[i][1] for i in sel])).to(f"cuda:{device}")
            y = eng.forward_wave(wave, win).cpu().numpy()
            for j, i in enumerate(sel):
                enhanced[i] = y[j]
    inf_scp, ref_scp = [], []
    for (wav_name, _, ref_path, n_clean), y in zip(items, enhanced):
        if y.shape[0] < n_clean:                                   # infer.py:98-102
            y = np.pad(y, (0, n_clean - y.shape[0]), mode="constant")
        elif y.shape[0] > n_clean:
            y = y[:n_clean]
        uid = wav_name.split(".wav")[0]
        enh_path = os.path.join(enh_dir, uid + "_enh.wav")
        write_wav_pcm16(enh_path, y)
        inf_scp.append((uid, enh_path))
        ref_scp.append((uid, ref_path))
    suffix = "" if world == 1 else f".rank{rank}"
    for fname, rows in (("inf.scp", inf_scp), ("ref.scp", ref_scp)):
        with open(os.path.join(enh_dir, fname + suffix), "w") as f:
            for uid, p in rows:
                f.write(f"{uid} {p}\n")
    return inf_scp, ref_scp


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--noisy-dir", required=True)
    ap.add_argument("--clean-dir", required=True)
    ap.add_argument("--enh-dir", required=True)
    ap.add_argument("--checkpoint", required=True, help="reference .tar checkpoint or raw fp32 blob (.f32)")
    ap.add_argument("-D", "--device", default=None, help="GPU index (default: LOCAL_RANK or 0)")
    ap.add_argument("--max-batch", type=int, default=64)
    a = ap.parse_args(argv)
    from .sharding import rank_world
    rank, local_rank, world = rank_world()
    dev = int(a.device) if a.device is not None else local_rank
    enhance_folder(a.noisy_dir, a.clean_dir, a.enh_dir, a.checkpoint, dev, a.max_batch, rank, world)


if __name__ == "__main__":
    main()
