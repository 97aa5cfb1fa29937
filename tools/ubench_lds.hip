// Microbenchmark: LDS read/write cost of the access patterns the kernels use (gfx950), 704 threads
// per workgroup (11 waves, as the model kernels), one workgroup per CU.
//   MODE 0  record layout:  lane (n,g) reads 16 B at (p0+n)*64 + 16g   (position-major 64-byte records)
//   MODE 1  plane layout:   lane (n,g) reads 16 B at g*PLANE + (p0+n)*16
//   MODE 2  record layout, XOR swizzle of the 16-byte chunk with (pos>>?)...
//   MODE 5  record layout with a 24-dword (96-byte) record stride: conflict-free for the b128 lane groups
//           {0-3,12-15,20-27},... of MI355X_MICROARCH.md (LDS section) at ANY base record
//   MODE 6/7/8  ds_write_b128 of a tile, record stride 16 / 24 / 20 dwords
//   MODE 3  scalar ds_read_b32, stride 12 floats across lanes (old ERB weight reads)
//   MODE 4  scalar ds_read_b32, consecutive
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NTHR = 704;
constexpr int PLANE = 16 * 35 * 4 + 4;   // floats per plane (+4: planes start 16 B apart mod 256 B)
template <int MODE>
__global__ __launch_bounds__(NTHR) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    for (int i = tid; i < 36 * 1024; i += NTHR) sm[i] = i * 1e-3f;
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    float accs = 0.f;
    int base;
    const int p0 = wave * 48;   // positions
    if (MODE == 0) base = (p0 + n) * 16 + 4 * g;
    else if (MODE == 1) base = g * PLANE + (p0 + n) * 4;
    else if (MODE == 2) base = (p0 + n) * 16 + 4 * ((g + (n >> 2)) & 3);
    else if (MODE == 5) base = (p0 + n) * 24 + 4 * g;
    else if (MODE == 6) base = (p0 + n) * 16 + 4 * g;
    else if (MODE == 7) base = (p0 + n) * 24 + 4 * g;
    else if (MODE == 8) base = (p0 + n) * 20 + 4 * g;
    else if (MODE == 3) base = lane * 12 + wave * 800;
    else base = lane + wave * 800;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(base));
        if (MODE >= 6) {
            f32x4 v = {1.f, 2.f, 3.f, (float)it};
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int rs = MODE == 6 ? 16 : MODE == 7 ? 24 : 20;
                const int off = (q % 3) * rs + (q / 3) * 35 * rs;
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(base * 4), "v"(v), "n"(off * 4) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE <= 2 || MODE == 5) {
            f32x4 v[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int rs = MODE == 5 ? 24 : 16;
                const int off = MODE == 1 ? (q % 3) * 4 + (q / 3) * 35 * 4 : (q % 3) * rs + (q / 3) * 35 * rs;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[q]) : "v"(base * 4), "n"(off * 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 9; q += 8) acc += v[q];
        } else {
#pragma unroll
            for (int q = 0; q < 12; ++q) accs += sm[base + q];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * NTHR + tid] = acc[0] + acc[1] + acc[2] + acc[3] + accs;
}
template <int MODE> float run(float* d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(NTHR), 150 * 1024, 0, d, iters); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(NTHR), 150 * 1024, 0, d, iters); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * NTHR * 4); const int iters = 4000;
    // LDS cycles per wave-instruction, per CU (11 waves share one LDS): ms * clk / (iters * instr * 11)
    auto cyc = [&](float ms, int instr) { return ms * 1e-3 * 2.0e9 / ((double)iters * instr * 11); };
    float t;
    t = run<0>(d, iters); printf("b128 record layout   : %.3f ms  %.1f cyc/wave-instr (at 2.0 GHz)\n", t, cyc(t, 9));
    t = run<1>(d, iters); printf("b128 plane layout    : %.3f ms  %.1f cyc/wave-instr\n", t, cyc(t, 9));
    t = run<2>(d, iters); printf("b128 record, swizzled: %.3f ms  %.1f cyc/wave-instr\n", t, cyc(t, 9));
    t = run<5>(d, iters); printf("b128 record stride 24: %.3f ms  %.1f cyc/wave-instr\n", t, cyc(t, 9));
    t = run<6>(d, iters); printf("write b128 stride 16 : %.3f ms  %.1f cyc/wave-instr\n", t, cyc(t, 9));
    t = run<7>(d, iters); printf("write b128 stride 24 : %.3f ms  %.1f cyc/wave-instr\n", t, cyc(t, 9));
    t = run<8>(d, iters); printf("write b128 stride 20 : %.3f ms  %.1f cyc/wave-instr\n", t, cyc(t, 9));
    t = run<3>(d, iters); printf("b32 stride 12        : %.3f ms  %.1f cyc/wave-instr\n", t, cyc(t, 12));
    t = run<4>(d, iters); printf("b32 consecutive      : %.3f ms  %.1f cyc/wave-instr\n", t, cyc(t, 12));
    return 0;
}
