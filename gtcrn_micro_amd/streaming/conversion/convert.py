"""Mirror of ``gtcrn_micro.streaming.conversion.convert.convert_to_stream`` (convert.py:7-56).

The reference remaps offline keys onto the streaming module's wrapper-prefixed keys and, for the
decoder's ``ConvTranspose2d`` weights, applies ``permute(1,0,2,3)`` + ``flip(-2,-1)`` so that a plain
``Conv2d`` over ``[cache | x]`` reproduces the transposed convolution.  Here the streaming model
shares the offline key names, and the permute/flip contract is honoured inside the packer
(csrc/pack.cpp: tap (kt,kf) of the dense 3x3 reads h[t-kt, f+1-kf]), so conversion is a checked copy.
"""


def convert_to_stream(stream_model, model) -> None:
    src = model.state_dict()
    dst = stream_model.state_dict()
    new = {}
    for key in dst.keys():
        cand = [key, key.replace("Conv1d.", ""), key.replace("Conv2d.", ""), key.replace(".deconv", ""),
                key.replace("ConvTranspose2d.", "")]
        hit = next((c for c in cand if c in src), None)
        if hit is None:
            raise ValueError("Key error!")
        new[key] = src[hit]
    stream_model.load_state_dict(new)
