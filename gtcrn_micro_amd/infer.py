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
    """16-bit mono PCM, the canonical 44-byte header -- byte for byte what ``scipy.io.wavfile.write`` (and libsndfile for
    the reference, infer.py:107) produce, written with one ``struct.pack`` and ``ndarray.tofile``: a folder of short clips
    spent more time in the generic writer's Python than in the kernels (tests/test_host_logic.py pins the bytes)."""
    import struct
    data = np.clip(np.rint(x * 32768.0), -32768, 32767).astype("<i2")
    nbytes = data.size * 2
    header = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + nbytes, b"WAVE", b"fmt ", 16, 1, 1, fs, fs * 2, 2, 16, b"data",
                         nbytes)
    with open(path, "wb") as f:
        f.write(header)
        data.tofile(f)


def load_params(checkpoint):
    """``.tar``/``.pt`` checkpoint of the reference (``ckpt["model"]``, train.py:200-216) or a raw fp32 blob."""
    if checkpoint.endswith((".f32", ".bin")):
        return np.fromfile(checkpoint, dtype=np.float32)
    import torch
    from .models.gtcrn_micro import state_dict_to_blob
    ck = torch.load(checkpoint, map_location="cpu", weights_only=False)
    return state_dict_to_blob(ck["model"] if "model" in ck else ck)


_READ_WAV_F32 = read_wav_f32      # (a test that swaps the generic reader for a failing one must not be bypassed by the fast path)

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


def _finish_clip(y_row, item, enh_dir):
    """Length-match one enhanced row to its clean reference (infer.py:98-102), write it, return its scp entry."""
    wav_name, _, ref_path, n_clean, L = item
    yj = y_row[:256 * (L // 256)]
    if yj.shape[0] < n_clean:
        yj = np.pad(yj, (0, n_clean - yj.shape[0]), mode="constant")
    elif yj.shape[0] > n_clean:
        yj = yj[:n_clean]
    uid = wav_name.split(".wav")[0]
    enh_path = os.path.join(enh_dir, uid + "_enh.wav")
    write_wav_pcm16(enh_path, yj)
    return uid, enh_path, ref_path


def _enhance_shard(noisy_dir, clean_dir, enh_dir, checkpoint, device=0, max_batch=64, rank=0, world=1,
                   barrier=None, pipeline=True, stats=None, stages=3, io_threads=4):
    """barrier: a callable all ranks call once their lists are written (multi-GPU runs; main() passes
    torch.distributed.barrier); rank 0 then merges the per-rank scp files.

    pipeline=True (default): the shard runs as a three-stage pipeline over `stages` staging slots, each a pinned host
    input buffer, a device input / output pair and a pinned host output buffer:
      reader thread   wav files -> float32 rows of slot k's pinned input buffer (zero tails)
      launch thread   (the caller's) H2D on a copy-in HIP stream, the six kernels on the compute stream, D2H on a
                      copy-out stream -- all asynchronous, ordered by events; batch k+1 uploads while k computes
      writer thread   waits for slot k's D2H event, length-matches and writes the 16-bit files, frees the slot
    (the reader and the writer each spread the clips of a batch over `io_threads` pool threads)
    so disk reads, PCIe both ways, the kernels and disk writes of different batches overlap.  The batches and the
    kernels are the ones of the serial form (pipeline=False: read -> pageable copy -> kernels -> copy back -> write, one
    batch after the other, what round 4 shipped), so every output file is bit-identical between the two.
    stats (optional dict) receives: clips, frames, wall_s, frames_per_s, gpu_busy_frac (device time of the kernel
    sequences / wall time, from events), h2d_bytes, d2h_bytes."""
    import queue
    import threading
    import time
    import warnings

    import torch
    from . import Engine, _lib
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
    fast = {}
    for wav_name in names[lo:hi]:
        path, ref_path, n_clean = _clip_info(noisy_dir, clean_dir, wav_name)
        fs, x = wavfile.read(path, mmap=True)
        if fs != 16000:
            raise AssertionError(f"{path}: sample rate {fs} != 16000 (resampling is not part of this path: the "
                                 "reference resamples with librosa, infer.py:54-57; resample the folder first)")
        if x.shape[0] < MIN_SAMPLES:
            warnings.warn(f"{path}: {x.shape[0]} samples < {MIN_SAMPLES}, cannot be reflect-padded: skipped")
            continue
        items.append((wav_name, path, ref_path, n_clean, int(x.shape[0])))
        # mono 16-bit PCM (what the reference's data are): pass 2 reads the samples straight from their byte offset
        fast[len(items) - 1] = int(x.offset) if (isinstance(x, np.memmap) and x.ndim == 1 and x.dtype == np.int16) else None
    # pass 2: batches of similar lengths, each through one launch sequence, written as soon as it is done
    order = sorted(range(len(items)), key=lambda i: items[i][4])
    batches = [order[k:k + max_batch] for k in range(0, len(order), max_batch)]
    rows = {}
    t_start = time.perf_counter()
    frames = sum(1 + items[i][4] // 256 for i in order)
    busy_ms = 0.0
    setup_s = 0.0
    nbytes = [0, 0]

    if not pipeline or not batches:
        for sel in batches:
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
            nbytes[0] += 4 * len(sel) * Lmax
            nbytes[1] += 4 * y.size
            for j, i in enumerate(sel):
                rows[i] = _finish_clip(y[j], items[i], enh_dir)
    else:
        # ---- staging slots: capacity = the largest batch (clips x longest clip) of this shard.  The samples cross the
        # host link as what they are on disk: 16-bit PCM in (a batch of mono 16-bit files is read straight into the pinned
        # buffer and widened to float32 on the DEVICE: int16 / 32768 is exact), 16-bit PCM out (rint(y * 32768) clipped to
        # the int16 range on the device -- numpy's rint and torch.round both round half to even -- so the host writes the
        # bytes it receives).  Half the link traffic, and no per-sample work on the host at all; a batch with any other kind
        # of file goes through the generic reader into float32 staging.
        cap = max(len(sel) * max(items[i][4] for i in sel) for sel in batches)
        cap = (cap + 7) & ~7                          # (the conversion kernels move 8 samples per lane)
        nslot = max(1, min(int(stages), len(batches)))
        with torch.cuda.device(device):
            slots = [{"hin": torch.empty(cap * 4, dtype=torch.uint8, pin_memory=True),
                      "hout": torch.empty(cap, dtype=torch.int16, pin_memory=True),
                      "din": torch.empty(cap * 4, dtype=torch.uint8, device=dev),
                      "dout": torch.empty(cap, dtype=torch.int16, device=dev)} for _ in range(nslot)]
            xf = torch.empty(cap, dtype=torch.float32, device=dev)      # the compute stream takes one batch at a time:
            yf = torch.empty(cap, dtype=torch.float32, device=dev)      # ONE float pair serves every slot
            s_in, s_cmp, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        setup_s = time.perf_counter() - t_start       # page-locking the staging buffers: a fixed cost per shard
        free = threading.Semaphore(nslot)
        q_read, q_write = queue.Queue(), queue.Queue()
        abort = threading.Event()
        errors = []
        generic = read_wav_f32 is not _READ_WAV_F32    # (a test swapped the generic reader: do not bypass it)

        def reader():
            try:
                for k, sel in enumerate(batches):
                    while not free.acquire(timeout=0.2):
                        if abort.is_set():
                            return
                    if abort.is_set():
                        return
                    lens = [items[i][4] for i in sel]
                    Lmax = max(lens)
                    pcm = (not generic) and all(fast.get(i) is not None for i in sel)
                    raw = slots[k % nslot]["hin"].numpy()
                    hin = (raw[:len(sel) * Lmax * 2].view(np.int16) if pcm else raw[:len(sel) * Lmax * 4].view(np.float32)
                           ).reshape(len(sel), Lmax)

                    def fill(j, hin=hin, sel=sel, lens=lens, pcm=pcm):
                        path = items[sel[j]][1]
                        if pcm:
                            with open(path, "rb", buffering=0) as f:
                                f.seek(fast[sel[j]])
                                got = f.readinto(memoryview(hin[j, :lens[j]]).cast("B"))
                            if got != 2 * lens[j]:
                                raise RuntimeError(f"{path} changed length while the folder was being enhanced")
                        else:
                            w = read_wav_f32(path)[1]
                            if len(w) != lens[j]:
                                raise RuntimeError(f"{path} changed length while the folder was being enhanced")
                            hin[j, :len(w)] = w
                        hin[j, lens[j]:] = 0
                    list(pool_r.map(fill, range(len(sel))))          # (file reads drop the GIL)
                    q_read.put((k, sel, lens, Lmax, pcm))
            except Exception as e:
                errors.append(e)
                abort.set()
            finally:
                q_read.put(None)

        def finish_i16(row, item):
            """The int16 form of _finish_clip: crop to the enhanced length, pad / crop to the clean length (infer.py:98-102),
            canonical header + the samples as received."""
            import struct
            wav_name, _, ref_path, n_clean, L = item
            n = min(256 * (L // 256), n_clean)
            uid = wav_name.split(".wav")[0]
            enh_path = os.path.join(enh_dir, uid + "_enh.wav")
            header = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + 2 * n_clean, b"WAVE", b"fmt ", 16, 1, 1, 16000, 32000, 2,
                                 16, b"data", 2 * n_clean)
            with open(enh_path, "wb") as f:
                f.write(header)
                row[:n].tofile(f)
                if n_clean > n:
                    f.write(bytes(2 * (n_clean - n)))
            return uid, enh_path, ref_path

        def writer():
            try:
                while True:
                    job = q_write.get()
                    if job is None:
                        return
                    k, sel, Lout, ev_done, ev_a, ev_b = job
                    ev_done.synchronize()
                    y = slots[k % nslot]["hout"].numpy()[:len(sel) * Lout].reshape(len(sel), Lout)
                    for i, row in zip(sel, pool_w.map(lambda j, y=y, sel=sel: finish_i16(y[j], items[sel[j]]),
                                                      range(len(sel)))):
                        rows[i] = row
                    busy.append(ev_a.elapsed_time(ev_b))
                    free.release()
            except Exception as e:
                errors.append(e)
                abort.set()

        busy = []
        from concurrent.futures import ThreadPoolExecutor
        pool_r = ThreadPoolExecutor(max(1, int(io_threads)), thread_name_prefix="gtcrn-folder-rd")
        pool_w = ThreadPoolExecutor(max(1, int(io_threads)), thread_name_prefix="gtcrn-folder-wr")
        th_r = threading.Thread(target=reader, name="gtcrn-folder-reader", daemon=True)
        th_w = threading.Thread(target=writer, name="gtcrn-folder-writer", daemon=True)
        th_r.start()
        th_w.start()
        try:
            with torch.cuda.device(device):
                while not abort.is_set():
                    try:
                        job = q_read.get(timeout=0.2)
                    except queue.Empty:
                        continue
                    if job is None:
                        break
                    k, sel, lens, Lmax, pcm = job
                    sl = slots[k % nslot]
                    n, Lout = len(sel), 256 * (Lmax // 256)
                    nb = n * Lmax * (2 if pcm else 4)
                    x = xf[:n * Lmax].view(n, Lmax)
                    y = yf[:n * Lout].view(n, Lout)
                    with torch.cuda.stream(s_in):
                        sl["din"][:nb].copy_(sl["hin"][:nb], non_blocking=True)
                        ev_in = torch.cuda.Event()
                        ev_in.record()
                    with torch.cuda.stream(s_cmp):
                        s_cmp.wait_event(ev_in)
                        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        ev_a.record()
                        if pcm:
                            # int16 -> float32 (s / 32768, exact), one pass (gtcrn_pcm16_to_f32); the count rounded up to the
                            # kernel's 8 samples per lane: the slack lies inside both buffers and nothing reads it
                            n8 = (n * Lmax + 7) & ~7
                            _lib.pcm16_to_f32(sl["din"][:2 * n8].view(torch.int16), out=xf[:n8])
                        else:
                            x.copy_(sl["din"][:nb].view(torch.float32).view(n, Lmax))
                        if min(lens) == Lmax:
                            eng.forward_wave(x, win, out=y)
                        else:
                            eng.forward_wave_var(x, lens, win, out=y)
                        # write_wav_pcm16's conversion on the device: rint(y * 32768) clipped to the int16 range, one pass
                        # (gtcrn_f32_to_pcm16; n * Lout is a multiple of 256)
                        _lib.f32_to_pcm16(yf[:n * Lout], out=sl["dout"][:n * Lout])
                        ev_b.record()
                    with torch.cuda.stream(s_out):
                        s_out.wait_event(ev_b)
                        sl["hout"][:n * Lout].copy_(sl["dout"][:n * Lout], non_blocking=True)
                        ev_done = torch.cuda.Event()
                        ev_done.record()
                    nbytes[0] += nb
                    nbytes[1] += 2 * n * Lout
                    q_write.put((k, sel, Lout, ev_done, ev_a, ev_b))
        except Exception as e:
            errors.append(e)
            abort.set()
        finally:
            q_write.put(None)
            th_w.join()
            if errors:
                abort.set()
            th_r.join()
            pool_r.shutdown(wait=True)
            pool_w.shutdown(wait=True)
            torch.cuda.synchronize(device)
        if errors:
            raise errors[0]
        busy_ms = float(sum(busy))
    wall = time.perf_counter() - t_start
    if stats is not None:
        stats.update({"clips": len(items), "batches": len(batches), "frames": int(frames), "wall_s": wall,
                      "setup_s": setup_s,
                      "frames_per_s": frames / wall if wall > 0 else 0.0,
                      "gpu_busy_frac": (busy_ms * 1e-3 / wall) if (pipeline and wall > 0) else None,
                      "h2d_bytes": nbytes[0], "d2h_bytes": nbytes[1], "pipeline": bool(pipeline)})
    inf_scp = [(rows[i][0], rows[i][1]) for i in range(len(items))]   # sorted file order, like the reference
    ref_scp = [(rows[i][0], rows[i][2]) for i in range(len(items))]
    suffix = "" if world == 1 else f".rank{rank}"
    for fname, lst in (("inf.scp", inf_scp), ("ref.scp", ref_scp)):
        with open(os.path.join(enh_dir, fname + suffix), "w") as f:
            for uid, p in lst:
                f.write(f"{uid} {p}\n")
    return inf_scp, ref_scp


def enhance_folder(noisy_dir, clean_dir, enh_dir, checkpoint, device=0, max_batch=64, rank=0, world=1, barrier=None,
                   agree=None, pipeline=True, stats=None, io_threads=4):
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
        result = _enhance_shard(noisy_dir, clean_dir, enh_dir, checkpoint, device, max_batch, rank, world, barrier,
                                pipeline=pipeline, stats=stats, io_threads=io_threads)
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
    ap.add_argument("--serial", action="store_true",
                    help="one batch after the other with pageable copies (the A/B of the pipelined default)")
    ap.add_argument("--stats", action="store_true", help="print this rank's throughput / GPU-busy figures as JSON")
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
        st = {} if a.stats else None
        enhance_folder(a.noisy_dir, a.clean_dir, a.enh_dir, a.checkpoint, dev, a.max_batch, rank, world, barrier, agree,
                       pipeline=not a.serial, stats=st)
        if st is not None:
            import json
            print(json.dumps({"rank": rank, **st}), flush=True)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
