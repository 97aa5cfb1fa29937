"""Mirror of ``gtcrn_micro.streaming.conversion.convert.convert_to_stream`` (convert.py:7-56).

Copies the offline weights into the streaming module: keys that differ only by the wrapper prefixes
``Conv1d.`` / ``Conv2d.`` / ``.deconv`` are remapped, and a ``ConvTranspose2d.``-prefixed weight is
``permute(1,0,2,3)`` + ``flip(-2,-1)`` so that a plain ``Conv2d`` over ``[cache | x]`` reproduces the
transposed convolution (:35-48).  For the fused models (``StreamGTCRNMicro`` here shares the offline key
names) this reduces to a checked copy; the permute/flip contract of the decoder's dense 3x3 is honoured
inside the packer (csrc/pack.cpp: tap (kt,kf) reads h[t-kt, f+1-kf]).  Unmatched keys raise
``ValueError("Key error!")`` like the reference (:54).
"""
import torch


def convert_to_stream(stream_model, model) -> None:
    src = model.state_dict()
    dst = stream_model.state_dict()
    new = {}
    for key in dst.keys():
        if key in src:
            new[key] = src[key]
        elif key.replace("Conv1d.", "") in src:
            new[key] = src[key.replace("Conv1d.", "")]
        elif key.replace("Conv2d.", "") in src:
            new[key] = src[key.replace("Conv2d.", "")]
        elif key.replace(".deconv", "") in src:
            new[key] = src[key.replace(".deconv", "")]
        elif key.replace("ConvTranspose2d.", "") in src:
            w = src[key.replace("ConvTranspose2d.", "")]
            if key.endswith("weight"):
                w = torch.flip(w.permute(1, 0, 2, 3).contiguous(), dims=[-2, -1])
                assert w.shape == dst[key].shape, (w.shape, dst[key].shape)
            new[key] = w
        else:
            raise ValueError("Key error!")
    stream_model.load_state_dict(new)
