// Microbenchmark: do v_mfma_f32_16x16x4_f32 and fp32 VALU FMAs overlap on one SIMD of gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_valu.hip -o /tmp/ub && /tmp/ub
// mode 0: MFMA only; 1: VALU only; 2: both in every wave (interleaved 1 MFMA : K VALU);
// 3: half of the waves MFMA only, other half VALU only (co-resident on the same SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int KV>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {1, 1, 1, 1}, acc2 = {2, 2, 2, 2}, acc3 = {3, 3, 3, 3};
    float a = 1.0f + lane * 1e-6f, b = 0.5f;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = q + lane;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int r = 0; r < KV; ++r)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaf(v[q], a, b);
        }
    }
    float s = acc0[0] + acc1[1] + acc2[2] + acc3[3];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int KV>
float run(float* d, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, KV>), dim3(256), dim3(512), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, KV>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    // per iteration and wave: 4 MFMA (= 4*32 = 128 MFMA-pipe cycles), 8*KV VALU FMAs
    float m = run<0, 1>(d, iters);
    printf("MFMA only (8 waves/CU, 2 per SIMD): %.3f ms -> %.1f cycles per MFMA per SIMD @2.4GHz\n", m,
           m * 1e-3 * 2.4e9 / (iters * 4.0 * 2));
    float v1 = run<1, 1>(d, iters), v2 = run<1, 2>(d, iters), v4 = run<1, 4>(d, iters);
    printf("VALU only KV=1: %.3f ms (%.2f cyc/instr/SIMD)  KV=2: %.3f  KV=4: %.3f (%.2f cyc/instr/SIMD)\n", v1,
           v1 * 1e-3 * 2.4e9 / (iters * 8.0 * 2), v2, v4, v4 * 1e-3 * 2.4e9 / (iters * 32.0 * 2));
    float b1 = run<2, 1>(d, iters), b2 = run<2, 2>(d, iters), b4 = run<2, 4>(d, iters);
    printf("both in every wave: KV=1 %.3f ms (sum %.3f, max %.3f)  KV=2 %.3f (sum %.3f)  KV=4 %.3f (sum %.3f, max %.3f)\n",
           b1, m + v1, m > v1 ? m : v1, b2, m + v2, b4, m + v4, m > v4 ? m : v4);
    float s1 = run<3, 1>(d, iters), s4 = run<3, 4>(d, iters), s8 = run<3, 8>(d, iters);
    printf("split waves (4 MFMA-only + 4 VALU-only per CU): KV=1 %.3f  KV=4 %.3f  KV=8 %.3f ms "
           "(MFMA-only half alone would take %.3f)\n", s1, s4, s8, m / 2);
    return 0;
}
