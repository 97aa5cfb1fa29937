#!/usr/bin/env python3
"""The reference's own known-answer vectors at FULL length: examples/gtcrn_micro/noisy{1..5}.wav -> enh{1..5}.wav
(five 31-second 16 kHz clips, T = 1 938 frames; produced by infer.py:48-107 with the shipped checkpoint).

Run only in the build container (the reference does not travel to the GPU box):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_examples.py

Writes examples_full.npz: `noisy` (5, 496000) int16 and `enh` (5, 495872) int16 -- the PCM samples of the ten wav
files, nothing else (data the reference's repository holds as test material; no source text).  Before writing, the
reference model is run on every clip (offline, infer.py's op sequence) and, for clip 1, frame by frame through its
StreamGTCRNMicro (gtcrn_micro_stream.py:618-635); how closely the reference reproduces its own files here (torch
build recorded) goes to MANIFEST_examples.json: that is the noise floor the HIP tests are held to.
"""
import json
import os
import sys
import types

import numpy as np
import torch
from scipy.io import wavfile

REF = "/root/reference"
sys.path.insert(0, REF)
sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))  # only used in __main__

from gtcrn_micro.models.gtcrn_micro import GTCRNMicro  # noqa: E402
from gtcrn_micro.streaming.gtcrn_micro_stream import StreamGTCRNMicro  # noqa: E402
from gtcrn_micro.streaming.conversion.convert import convert_to_stream  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
EX = os.path.join(REF, "gtcrn_micro/examples/gtcrn_micro")


def main():
    torch.set_num_threads(8)
    ck = torch.load(os.path.join(REF, "gtcrn_micro/ckpts/best_model_dns3.tar"), map_location="cpu", weights_only=False)
    model = GTCRNMicro().eval()
    model.load_state_dict(ck["model"])
    win = torch.hann_window(512).pow(0.5)                                   # infer.py:65
    noisy, enh, meta = [], [], {"torch": torch.__version__, "numpy": np.__version__, "clips": []}
    for i in range(1, 6):
        fs, x = wavfile.read(os.path.join(EX, f"noisy{i}.wav"))
        fs2, y = wavfile.read(os.path.join(EX, f"enh{i}.wav"))
        assert fs == fs2 == 16000 and x.dtype == np.int16 and y.dtype == np.int16 and x.ndim == 1
        noisy.append(x)
        enh.append(y)
        xf = torch.from_numpy(x.astype(np.float32) / 32768.0)
        with torch.inference_mode():
            spec = torch.stft(xf, 512, 256, 512, win, return_complex=False)[None]
            out = model(spec)[0]
            yr = torch.istft(torch.view_as_complex(out.contiguous()), 512, 256, 512, win).numpy()
        assert len(yr) == len(y), (len(yr), len(y))
        meta["clips"].append({"clip": i, "samples_noisy": int(len(x)), "samples_enh": int(len(y)),
                              "frames": int(spec.shape[2]),
                              "reference_offline_vs_file_max_lsb": float(np.abs(yr * 32768.0 - y).max())})
        if i == 1:   # the reference's own streaming loop on clip 1
            sm = StreamGTCRNMicro().eval()
            convert_to_stream(sm, model)
            conv_cache = torch.zeros(2, 1, 16, 6, 33)
            tra_cache = torch.zeros(2, 3, 1, 8, 2)
            tcn_cache = [[torch.zeros(1, 16, 2 * d, 33) for d in (1, 2, 4, 8)] for _ in range(2)]
            ys = []
            with torch.no_grad():
                for t in range(spec.shape[2]):
                    yt, conv_cache, tra_cache, tcn_cache = sm(spec[:, :, t:t + 1], conv_cache, tra_cache, tcn_cache)
                    ys.append(yt)
            ysp = torch.cat(ys, 2)[0]
            ys_w = torch.istft(torch.view_as_complex(ysp.contiguous()), 512, 256, 512, win).numpy()
            meta["clips"][-1]["reference_streamed_vs_file_max_lsb"] = float(np.abs(ys_w * 32768.0 - y).max())
            meta["clips"][-1]["reference_streamed_vs_offline_spec_maxabs"] = float((ysp - out).abs().max())
    noisy, enh = np.stack(noisy), np.stack(enh)
    np.savez_compressed(os.path.join(OUT, "examples_full.npz"), noisy=noisy, enh=enh)
    json.dump(meta, open(os.path.join(OUT, "MANIFEST_examples.json"), "w"), indent=1)
    print(json.dumps(meta, indent=1))
    print("examples_full.npz:", os.path.getsize(os.path.join(OUT, "examples_full.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
