// TEST INFRASTRUCTURE: the host-side packer (csrc/pack.cpp: BN folding, slot renaming, store tables) under
// AddressSanitizer / UBSan.  Usage: asan_pack_driver <params.f32>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "layout.h"
#include "pack.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<float> p(gtl::NPARAM);
    if (std::fread(p.data(), sizeof(float), p.size(), f) != p.size()) return 2;
    std::fclose(f);
    const auto& tab = gtcrn::param_table();
    if ((int)tab.size() != gtl::NTENSORS || tab.back().offset + tab.back().numel != gtl::NPARAM) return 3;
    std::vector<float> F(gtl::P_FLOATS);
    std::vector<int> I(gtl::P_INTS);
    std::string err;
    if (gtcrn::pack_params(p.data(), (long)p.size(), F.data(), I.data(), err) != 0) return 4;
    if (gtcrn::pack_params(p.data(), (long)p.size() - 1, F.data(), I.data(), err) == 0) return 5;   // wrong size rejected
    double acc = 0.0;
    for (float v : F) acc += std::fabs((double)v);
    float w[512];
    gtcrn::make_window(0, w);
    gtcrn::make_window(1, w);
    if (!std::isfinite(acc)) return 6;
    std::printf("asan_pack_driver ok %.6f\n", acc);
    return 0;
}
