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

void make_window(int kind, float* w512);

}  // namespace gtcrn
