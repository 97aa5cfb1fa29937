// Read-only two-tensor streaming reduction (the shape of the BatchNorm backward's first pass: da and y of one
// unit, 2 x 271 MB at B = 512) in several access layouts, to find what keeps the pass at ~3.4 TB/s when an in-order
// sweep reads at 6 TB/s (MI355X_MICROARCH.md).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_reduce_bw.hip -o /tmp/ub_reduce && /tmp/ub_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// LAYOUT 0: grid-stride (thread i, i + G*NT, ...), U strides issued together
// LAYOUT 1: every workgroup owns one contiguous range; threads sweep it NT units at a time, U trips issued together
// LAYOUT 2: every WAVE owns one contiguous range (64 units per trip)
template <int NT, int U, int LAYOUT, bool NTL>
__global__ __launch_bounds__(NT) void k_red(const f32x4* __restrict__ a, const f32x4* __restrict__ b, long units,
                                           float* __restrict__ out) {
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    long i0, step, end;
    if (LAYOUT == 0) {
        i0 = (long)blockIdx.x * NT + threadIdx.x; step = (long)gridDim.x * NT; end = units;
    } else if (LAYOUT == 1) {
        const long per = (units + gridDim.x - 1) / gridDim.x;
        i0 = blockIdx.x * per + threadIdx.x; step = NT; end = (blockIdx.x + 1) * per < units ? (blockIdx.x + 1) * per : units;
    } else {
        const long nw = (long)gridDim.x * (NT / 64), w = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
        const long per = (units + nw - 1) / nw;
        i0 = w * per + (threadIdx.x & 63); step = 64; end = (w + 1) * per < units ? (w + 1) * per : units;
    }
    for (; i0 < end; i0 += U * step) {
        f32x4 x[U], g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * step < end ? i0 + u * step : i0;
            if (NTL) { x[u] = __builtin_nontemporal_load(a + i); g[u] = __builtin_nontemporal_load(b + i); }
            else { x[u] = a[i]; g[u] = b[i]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * step >= end) break;
            s1 += g[u];
            s2 += g[u] * x[u];
        }
    }
    __shared__ float sh[NT];
    float t = s1[0] + s1[1] + s1[2] + s1[3] + s2[0] + s2[1] + s2[2] + s2[3];
    sh[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float q = 0.f;
        for (int l = 0; l < NT; ++l) q += sh[l];
        out[blockIdx.x] = q;
    }
}

// rotating over NPAIR buffer pairs (3.3 GB): nothing of a pair is left in the 256 MB Infinity Cache when its turn comes
// again; with WRITER the second tensor of the pair is (re)written by a streaming kernel right before it is read, as the
// gradient `da` is in the train step
__global__ __launch_bounds__(256) void k_fill(f32x4* __restrict__ b, long units, float v) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < units; i += (long)gridDim.x * 256) b[i] = f32x4{v, v, v, v};
}
constexpr int NPAIR = 6;
template <int NT, int U, int LAYOUT, bool NTL>
static int run_rot(const char* name, int grid, f32x4* const* a, f32x4* const* b, long units, float* out, bool writer) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double us = 0.0;
    const int R = 18;
    for (int i = 0; i < R + 3; ++i) {
        const int k = i % NPAIR;
        if (writer) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, b[k], units, 0.f);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_red<NT, U, LAYOUT, NTL>), dim3(grid), dim3(NT), 0, 0, a[k], b[k], units, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 3) us += ms * 1e3 / R;
    }
    printf("%-44s grid %5d  %7.1f us  %5.2f TB/s  (rotating%s)\n", name, grid, us, 2.0 * units * 16 / us * 1e-6,
           writer ? ", second tensor just written" : "");
    return 0;
}

template <int NT, int U, int LAYOUT, bool NTL>
static int run(const char* name, int grid, const f32x4* a, const f32x4* b, long units, float* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_red<NT, U, LAYOUT, NTL>), dim3(grid), dim3(NT), 0, 0, a, b, units, out);
    CK(hipEventRecord(e0));
    const int R = 20;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL((k_red<NT, U, LAYOUT, NTL>), dim3(grid), dim3(NT), 0, 0, a, b, units, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / R;
    printf("%-44s grid %5d  %7.1f us  %5.2f TB/s\n", name, grid, us, 2.0 * units * 16 / us * 1e-6);
    return 0;
}

int main() {
    const long units = 512L * 251 * 33 * 4;     // 16-byte units of one [B*T*33][16] fp32 tensor
    f32x4 *a, *b;
    float* out;
    CK(hipMalloc(&a, units * 16)); CK(hipMalloc(&b, units * 16)); CK(hipMalloc(&out, 65536 * 4));
    CK(hipMemset(a, 0, units * 16)); CK(hipMemset(b, 0, units * 16));
    printf("two tensors of %.1f MB\n", units * 16 / 1e6);
    for (int grid : {1024, 2048, 4096}) run<256, 4, 0, false>("grid-stride NT256 U4 (current)", grid, a, b, units, out);
    for (int grid : {1024, 2048}) run<256, 8, 0, false>("grid-stride NT256 U8", grid, a, b, units, out);
    for (int grid : {1024, 2048}) run<256, 2, 0, false>("grid-stride NT256 U2", grid, a, b, units, out);
    for (int grid : {1024, 2048}) run<256, 1, 0, false>("grid-stride NT256 U1", grid, a, b, units, out);
    for (int grid : {512, 1024}) run<512, 4, 0, false>("grid-stride NT512 U4", grid, a, b, units, out);
    for (int grid : {256, 512}) run<1024, 4, 0, false>("grid-stride NT1024 U4", grid, a, b, units, out);
    for (int grid : {1024, 2048}) run<256, 4, 0, true>("grid-stride NT256 U4 nontemporal", grid, a, b, units, out);
    for (int grid : {1000, 1021}) run<256, 4, 0, false>("grid-stride NT256 U4 odd grid", grid, a, b, units, out);
    for (int grid : {1024, 2048, 4096}) run<256, 4, 1, false>("WG-contiguous NT256 U4", grid, a, b, units, out);
    for (int grid : {1024, 2048}) run<256, 8, 1, false>("WG-contiguous NT256 U8", grid, a, b, units, out);
    for (int grid : {1024, 2048}) run<256, 4, 1, true>("WG-contiguous NT256 U4 nontemporal", grid, a, b, units, out);
    for (int grid : {1024, 2048}) run<256, 4, 2, false>("wave-contiguous NT256 U4", grid, a, b, units, out);
    for (int grid : {1024, 2048}) run<256, 8, 2, false>("wave-contiguous NT256 U8", grid, a, b, units, out);
    f32x4 *ra[NPAIR], *rb[NPAIR];
    for (int k = 0; k < NPAIR; ++k) {
        CK(hipMalloc(&ra[k], units * 16)); CK(hipMalloc(&rb[k], units * 16));
        CK(hipMemset(ra[k], 0, units * 16)); CK(hipMemset(rb[k], 0, units * 16));
    }
    for (int w = 0; w < 2; ++w) {
        run_rot<256, 4, 0, false>("grid-stride NT256 U4", 1024, ra, rb, units, out, w);
        run_rot<256, 4, 0, false>("grid-stride NT256 U4", 2048, ra, rb, units, out, w);
        run_rot<256, 8, 0, false>("grid-stride NT256 U8", 1024, ra, rb, units, out, w);
        run_rot<256, 4, 0, true>("grid-stride NT256 U4 nontemporal", 1024, ra, rb, units, out, w);
        run_rot<256, 4, 1, false>("WG-contiguous NT256 U4", 1024, ra, rb, units, out, w);
        run_rot<1024, 4, 0, false>("grid-stride NT1024 U4", 256, ra, rb, units, out, w);
    }
    return 0;
}
