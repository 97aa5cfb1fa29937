"""Bulk offline enhancement on MI355X: the counterpart of the reference's ``gtcrn_micro/infer.py``.

Same contract as ``infer.py:26-119``: every ``*.wav`` of the noisy folder (sorted) is enhanced with the
checkpoint, length-matched to its clean reference (zero-pad / crop, ``:98-102``), written as
``<uid>_enh.wav`` and listed in ``inf.scp`` / ``ref.scp`` (``:113-119``).  Differences, all on the host side:

* clips of ANY lengths are packed (sorted by length, so a batch holds similar ones) into full batches and go through
  one fused wave->wave launch sequence with per-clip lengths (``gtcrn_forward_wave_var``: STFT -> model -> iSTFT on
  the GPU, each clip bit-identical to its own single-clip result); every batch is written as soon as it is enhanced;
* several GPUs shard the sorted file list (one process per GPU, no collective; ``sharding.shard_range``); rank 0
  merges the per-rank lists into the single ``inf.scp`` / ``ref.scp`` the reference's eval expects;
* a clip shorter than 257 samples cannot be reflect-padded by 256 (``torch.stft`` raises there too): it is skipped
  with a warning instead of aborting the folder;
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
    elif x.dtype == np.uint8:                      # 8-bit PCM is unsigned with an offset of 128
        x = (x.astype(np.float32) - 128.0) / 128.0
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


MIN_SAMPLES = 257      # reflect padding by 256 needs more than 256 samples (torch.stft raises below that, infer.py:60)


def _clip_info(noisy_dir, clean_dir, wav_name):
    """(path, reference path, clean length) of one noisy clip; the checks of infer.py:52-96."""
    from scipy.io import wavfile
    path = os.path.join(noisy_dir, wav_name)
    fileid = extract_fileid(path)
    if fileid is None:
        raise RuntimeError(f"Unable to extract: {path}")
    ref_path = os.path.join(clean_dir, f"clean_fileid_{fileid}.wav")
    if not os.path.exists(ref_path):
        raise FileNotFoundError(f"Clean file not found for clean_fileid_{fileid}.wav, fileid={fileid}:\n {ref_path}")
    fs_c, clean = wavfile.read(ref_path, mmap=True)
    if fs_c != 16000:
        raise AssertionError(f"{ref_path}: sample rate {fs_c} != 16000")
    return path, ref_path, int(clean.shape[0])


def merge_scp(enh_dir, world):
    """inf.scp.rank{r} / ref.scp.rank{r} -> inf.scp / ref.scp in rank order (= sorted file order)."""
    for fname in ("inf.scp", "ref.scp"):
        with open(os.path.join(enh_dir, fname), "w") as out:
            for r in range(world):
                with open(os.path.join(enh_dir, f"{fname}.rank{r}")) as f:
                    out.write(f.read())


def _enhance_shard(noisy_dir, clean_dir, enh_dir, checkpoint, device=0, max_batch=64, rank=0, world=1,
                   barrier=None):
    """barrier: a callable all ranks call once their lists are written (multi-GPU runs; main() passes
    torch.distributed.barrier); rank 0 then merges the per-rank scp files."""
    import warnings

    import torch
    from . import Engine
    from .sharding import shard_range

    os.makedirs(enh_dir, exist_ok=True)
    eng = Engine(load_params(checkpoint), device)
    dev = f"cuda:{device}"
    win = torch.hann_window(512).pow(0.5).to(dev)                   # infer.py:65
    names = sorted(f for f in os.listdir(noisy_dir) if f.endswith("wav"))
    lo, hi = shard_range(len(names), world, rank)
    # pass 1 (headers only): pair every clip with its reference, learn the lengths
    from scipy.io import wavfile
    items = []
    for wav_name in names[lo:hi]:
        path, ref_path, n_clean = _clip_info(noisy_dir, clean_dir, wav_name)
        fs, x = wavfile.read(path, mmap=True)
        if fs != 16000:
            raise AssertionError(f"{path}: sample rate {fs} != 16000 (resampling is not part of this path)")
        if x.shape[0] < MIN_SAMPLES:
            warnings.warn(f"{path}: {x.shape[0]} samples < {MIN_SAMPLES}, cannot be reflect-padded: skipped")
            continue
        items.append((wav_name, path, ref_path, n_clean, int(x.shape[0])))
    # pass 2: batches of similar lengths, each through one launch sequence, written as soon as it is done
    order = sorted(range(len(items)), key=lambda i: items[i][4])
    rows = {}
    for k in range(0, len(order), max_batch):
        sel = order[k:k + max_batch]
        waves = [read_wav_f32(items[i][1])[1] for i in sel]
        lens = [len(w) for w in waves]
        Lmax = max(lens)
        if min(lens) == Lmax:
            y = eng.forward_wave(torch.from_numpy(np.stack(waves)).to(dev), win).cpu().numpy()
        else:
            host = np.zeros((len(sel), Lmax), np.float32)
            for j, w in enumerate(waves):
                host[j, :len(w)] = w
            y = eng.forward_wave_var(torch.from_numpy(host).to(dev), lens, win).cpu().numpy()
        for j, i in enumerate(sel):
            wav_name, _, ref_path, n_clean, L = items[i]
            yj = y[j, :256 * (L // 256)]
            if yj.shape[0] < n_clean:                               # infer.py:98-102
                yj = np.pad(yj, (0, n_clean - yj.shape[0]), mode="constant")
            elif yj.shape[0] > n_clean:
                yj = yj[:n_clean]
            uid = wav_name.split(".wav")[0]
            enh_path = os.path.join(enh_dir, uid + "_enh.wav")
            write_wav_pcm16(enh_path, yj)
            rows[i] = (uid, enh_path, ref_path)
    inf_scp = [(rows[i][0], rows[i][1]) for i in range(len(items))]   # sorted file order, like the reference
    ref_scp = [(rows[i][0], rows[i][2]) for i in range(len(items))]
    suffix = "" if world == 1 else f".rank{rank}"
    for fname, lst in (("inf.scp", inf_scp), ("ref.scp", ref_scp)):
        with open(os.path.join(enh_dir, fname + suffix), "w") as f:
            for uid, p in lst:
                f.write(f"{uid} {p}\n")
    return inf_scp, ref_scp


def enhance_folder(noisy_dir, clean_dir, enh_dir, checkpoint, device=0, max_batch=64, rank=0, world=1, barrier=None,
                   agree=None):
    """Counterpart of infer.py:26-119 for a folder (see ``_enhance_shard``): every rank enhances its contiguous shard of
    the sorted file list, then rank 0 merges the per-rank scp lists.

    A failure on ONE rank (a clip with the wrong sample rate, a missing reference, an I/O error) must not leave the
    others waiting in a collective: the per-rank work runs inside try/finally, the ranks then exchange a failure flag
    -- ``agree(ok) -> bool`` (True only if every rank succeeded; ``main`` passes an all-reduce), or failing that the
    plain ``barrier`` -- and only a run in which every rank succeeded is merged.  The failing rank re-raises its own
    error; the others raise a RuntimeError naming the situation, so every process exits non-zero."""
    err = None
    result = None
    try:
        result = _enhance_shard(noisy_dir, clean_dir, enh_dir, checkpoint, device, max_batch, rank, world, barrier)
    except Exception as e:           # re-raised below, after the exchange.  NOT BaseException: a KeyboardInterrupt or
        err = e                      # SystemExit must leave at once instead of waiting in a collective first
    all_ok = err is None
    if world > 1:
        if agree is not None:
            all_ok = bool(agree(err is None))
        elif barrier is not None:
            barrier()
    if err is not None:
        raise err
    if not all_ok:
        raise RuntimeError("enhance_folder: another rank failed; its shard is incomplete, the scp lists were not merged")
    if world > 1 and rank == 0 and (agree is not None or barrier is not None):
        merge_scp(enh_dir, world)
    return result


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
    if world > 1 and a.device is not None:
        # init_distributed binds RCCL to cuda:LOCAL_RANK; one explicit device for every rank would put all of them
        # on the same GPU while their communicators sit on different ones
        ap.error("--device cannot be combined with a multi-process launch (WORLD_SIZE > 1): each rank uses cuda:LOCAL_RANK")
    dev = int(a.device) if a.device is not None else local_rank
    barrier = agree = None
    if world > 1:
        import torch
        import torch.distributed as dist
        from .sharding import init_distributed
        init_distributed(None)
        barrier = dist.barrier

        def agree(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                                device=f"cuda:{local_rank}" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())
    try:
        enhance_folder(a.noisy_dir, a.clean_dir, a.enh_dir, a.checkpoint, dev, a.max_batch, rank, world, barrier, agree)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
