"""Mirror of ``gtcrn_micro.streaming.conversion.convert.convert_to_stream`` (convert.py:7-56).

Contract: every key of the streaming module's state_dict takes the offline tensor of the same name once the
streaming wrapper's infix is removed; a ``ConvTranspose2d.`` weight is additionally turned into the weight of
the equivalent plain ``Conv2d`` over ``[cache | x]``: W'[o,i,a,b] = W[i,o,kT-1-a,kF-1-b] (:35-48).  A key with
no counterpart raises ``ValueError("Key error!")`` (:54).  For the fused models (``StreamGTCRNMicro`` shares
the offline key names) this is a checked copy; the same permute/flip contract for the decoder's dense 3x3
lives in the packer (csrc/pack.cpp: tap (kt,kf) reads h[t-kt, f+1-kf]).
"""


def _as_conv2d_weight(w, like):
    w = w.permute(1, 0, 2, 3).flip(-2, -1).contiguous()
    assert w.shape == like.shape, (w.shape, like.shape)
    return w


# (infix the streaming wrapper adds to the key, transform of a ``weight`` tensor or None)
_WRAPPER_INFIXES = (
    ("", None),
    ("Conv1d.", None),
    ("Conv2d.", None),
    (".deconv", None),
    ("ConvTranspose2d.", _as_conv2d_weight),
)


def _lookup(key, like, src):
    for infix, weight_fn in _WRAPPER_INFIXES:
        name = key.replace(infix, "") if infix else key
        if name in src and (infix == "" or infix in key):
            t = src[name]
            return weight_fn(t, like) if weight_fn and key.endswith("weight") else t
    raise ValueError("Key error!")


def convert_to_stream(stream_model, model) -> None:
    src = model.state_dict()
    dst = stream_model.state_dict()
    stream_model.load_state_dict({key: _lookup(key, like, src) for key, like in dst.items()})
