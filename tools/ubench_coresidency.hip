// Microbenchmark: when do TWO workgroups of six waves share a CU on gfx950?  (the pair form of the offline encoder)
// 512 workgroups of 384 threads spin for a fixed number of s_memtime ticks; a launch takes one spin when two workgroups
// fit on every one of the 256 CUs and two spins when they do not.  Swept: dynamic LDS bytes per workgroup and the
// VGPR allocation (forced with an inline-asm touch of the highest register).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_coresidency.hip -o /tmp/ubench_coresidency && /tmp/ubench_coresidency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int VG>
__global__ __launch_bounds__(384) void spin(unsigned long long ticks, int* out) {
    extern __shared__ int sm[];
    if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (VG == 136) asm volatile("v_mov_b32 v135, 0" ::: "v135");
    if (VG == 144) asm volatile("v_mov_b32 v143, 0" ::: "v143");
    if (VG == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    sm[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (out && sm[threadIdx.x] == -1) out[0] = 1;
}
template <int VG>
static float run(int lds, int nwg) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(spin<VG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    spin<VG><<<nwg, 384, lds>>>(400000ull, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(a);
    spin<VG><<<nwg, 384, lds>>>(400000ull, nullptr);      // s_memtime ticks at ~1.7 GHz here: ~0.24 ms per spin
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    const int ldss[] = {65536, 77264, 79872, 80896, 80976, 81408, 81920, 82432};
    printf("512 workgroups x 384 threads, one spin = the 256-workgroup time; ms per launch\n");
    printf("%8s %10s %10s %10s %10s %10s\n", "LDS B", "256wg/128v", "128 vgpr", "136 vgpr", "144 vgpr", "168 vgpr");
    for (int lds : ldss)
        printf("%8d %10.3f %10.3f %10.3f %10.3f %10.3f\n", lds, run<128>(lds, 256), run<128>(lds, 512), run<136>(lds, 512),
               run<144>(lds, 512), run<168>(lds, 512));
    return 0;
}
