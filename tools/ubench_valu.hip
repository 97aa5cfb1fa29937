// Microbenchmark: issue cost per SIMD of the VALU forms the kernels use (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-6f, b = 0.5f;
    f32x2 p[8]; float v[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) { p[q] = f32x2{(float)q + lane, (float)q - lane}; }
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = q + lane * 0.5f;
    const f32x2 pa = {a, a}, pb = {b, b};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // 16 scalar fma
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = fmaf(v[q], a, b);
        } else if (MODE == 1) {     // 8 packed fma (= 16 fma)
#pragma unroll
            for (int q = 0; q < 8; ++q) p[q] = __builtin_elementwise_fma(p[q], pa, pb);
        } else if (MODE == 2) {     // 16 min
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = fminf(v[q], a) + 0.0f * b;
        } else if (MODE == 3) {     // 16 dpp row_ror moves + add
#pragma unroll
            for (int q = 0; q < 16; ++q)
                v[q] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[q]), 0x121, 0xf, 0xf, false));
        } else if (MODE == 4) {     // 8 packed mul + 8 packed add
#pragma unroll
            for (int q = 0; q < 8; ++q) { p[q] = p[q] * pa; p[q] = p[q] + pb; }
        }
    }
    float s = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += p[q][0] + p[q][1];
#pragma unroll
    for (int q = 0; q < 16; ++q) s += v[q];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE> float run(float* d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, d, iters); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, d, iters); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4); const int iters = 20000;
    const double c = 1e-3 * 2.4e9 / (iters * 2.0);   // cycles per loop iteration per SIMD (2 waves/SIMD)
    printf("16 v_fma_f32      : %.3f ms  %.1f cyc/iter\n", run<0>(d, iters), run<0>(d, iters) * c);
    printf("8 v_pk_fma_f32    : %.3f ms  %.1f cyc/iter\n", run<1>(d, iters), run<1>(d, iters) * c);
    printf("16 v_min(+fma)    : %.3f ms  %.1f cyc/iter\n", run<2>(d, iters), run<2>(d, iters) * c);
    printf("16 dpp ror + add  : %.3f ms  %.1f cyc/iter\n", run<3>(d, iters), run<3>(d, iters) * c);
    printf("8 pk_mul + 8 pk_add: %.3f ms  %.1f cyc/iter\n", run<4>(d, iters), run<4>(d, iters) * c);
    return 0;
}
