// pack.h -- host-side parameter table and slot-space packer (see pack.cpp, layout.h).
#pragma once
#include <string>
#include <vector>

namespace gtcrn {

struct ParamInfo {
    std::string name;
    long numel;
    long offset;
};

// canonical order = reference state_dict minus num_batches_tracked (342 tensors, 44 938 floats)
const std::vector<ParamInfo>& param_table();

// F: gtl::P_FLOATS floats, I: gtl::P_INTS ints.  Returns 0 or -1 with err set.
int pack_params(const float* params, long n, float* F, int* I, std::string& err);

// int8-weight variant: in place on the packed float buffer, every conv / linear weight -> fp16(int8 * per-output-
// channel scale) (scripts/onnx2tf.sh:50-64: -oiqt -qt per-channel).  round_to_half: float -> binary16 (RNE) -> float.
void quantize_packed(float* F);
float round_to_half(float x);

void make_window(int kind, float* w512);

}  // namespace gtcrn
