// Microbenchmark: what does cold straight-line code cost on gfx950?  The single-launch streaming step is ~36-52 KB of
// straight-line code that every workgroup walks exactly ONCE (no loops: every phase is unrolled), at ~18 shader cycles per
// instruction -- far below what its dependency chains explain.  Hypothesis: instruction fetch.  Here: the same dynamic
// instruction count (independent v_fma chains, nothing else) either as ONE straight-line block of KB kilobytes or as a
// loop over a 2 KB body; one workgroup of NW waves per CU, one pass per workgroup, cycles from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_icache.hip -o /tmp/ubench_icache && /tmp/ubench_icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// 256 instructions x 8 bytes (VOP3) = 2 KB; four independent chains
#define BODY_2K                                                                   \
    asm volatile(".rept 64\n\t"                                                   \
                 "v_fma_f32 %0, %0, %4, %5\n\t"                                   \
                 "v_fma_f32 %1, %1, %4, %5\n\t"                                   \
                 "v_fma_f32 %2, %2, %4, %5\n\t"                                   \
                 "v_fma_f32 %3, %3, %4, %5\n\t"                                   \
                 ".endr"                                                          \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                             \
                 : "v"(m), "v"(k));
#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)

template <int KB, bool LOOP>
__global__ void k(float* out, unsigned long long* cyc, float m, float kk) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    const float k = kk;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (LOOP) {
#pragma unroll 1
        for (int i = 0; i < KB / 2; ++i) { BODY_2K }
    } else {
        if constexpr (KB == 64) { R32(BODY_2K) }
        else if constexpr (KB == 32) { R16(BODY_2K) }
        else if constexpr (KB == 16) { R8(BODY_2K) }
        else { R4(BODY_2K) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KB, bool LOOP>
void run(int nwg, int nw, float* out, unsigned long long* cyc) {
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<KB, LOOP>), dim3(nwg), dim3(nw * 64), 0, 0, out, cyc, 0.999f, 0.001f);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(nwg * nw);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2], mx = (double)h.back(), mn = (double)h.front();
    const int instr = KB * 128;
    printf("%-9s %2d KB  %4d wg x %2d waves: memtime ticks min %8.0f med %8.0f max %8.0f  -> %.2f ticks / instruction, %.1f ticks / 64 B line\n",
           LOOP ? "loop" : "straight", KB, nwg, nw, mn, med, mx, med / instr, med / (KB * 16));
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4 * 4096 * 1024);
    hipMalloc(&cyc, 8 * 4096 * 16);
    // s_memtime runs at a fixed 100 MHz on this part; the kernels' phase profile uses the same unit
    for (int nw : {1, 8, 11}) {
        for (int nwg : {256, 2048}) {
            run<8, false>(nwg, nw, out, cyc);
            run<16, false>(nwg, nw, out, cyc);
            run<32, false>(nwg, nw, out, cyc);
            run<64, false>(nwg, nw, out, cyc);
            run<64, true>(nwg, nw, out, cyc);
        }
    }
    return 0;
}
