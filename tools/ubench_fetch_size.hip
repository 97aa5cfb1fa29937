// ubench_fetch_size.hip -- calibrates rocprofv3's FETCH_SIZE (and WRITE_SIZE) for the access widths this repo's kernels
// use.  MI355X_MICROARCH.md (HBM): "On gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming
// read (16 B/lane) ... Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern."
// The 16-bit storage modes of the train step read 8 bytes per lane (four bf16), k_front / k_istft / k_stft read 4 and 8;
// their traffic figures were quoted as a band (FETCH x 1 ... FETCH x 2) in round 4.
//
// Every kernel below streams the SAME 1 GiB buffer (four times the Infinity Cache, so nothing is served on-die) once, each
// lane taking W contiguous bytes of a fully coalesced wave access; grid-stride over 2048 workgroups of 256 threads.
//   rd<W, NT>   read-only (sum kept alive through one store per workgroup), plain or nontemporal loads
//   cp<W>       read + write of the same width (WRITE_SIZE calibration)
// Run:   hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch_size.hip -o /tmp/ub_fetch
//        /tmp/ub_fetch                                            (GB/s per variant, HIP events)
//        rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out_f -- /tmp/ub_fetch 1
//        rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out_w -- /tmp/ub_fetch 1
// tools/fetch_calibration.py turns the two counter files into bytes-per-counted-KB factors per width.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <class V, bool NT>
__device__ __forceinline__ V ld(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <class V> __device__ __forceinline__ float fold(V v);
template <> __device__ __forceinline__ float fold<float>(float v) { return v; }
template <> __device__ __forceinline__ float fold<f32x2>(f32x2 v) { return v[0] + v[1]; }
template <> __device__ __forceinline__ float fold<f32x4>(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

template <class V, bool NT>
__global__ __launch_bounds__(256) void rd(const V* __restrict__ src, long n, float* __restrict__ out) {
    float s = 0.f;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += 4 * stride) {     // four loads in flight
        V v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld<V, NT>(src + (i + u * stride < n ? i + u * stride : i));
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (i + u * stride < n) ? fold<V>(v[u]) : 0.f;
    }
    // one value per wave leaves the kernel: the loads cannot be optimised away, the stores are 8 KB in all
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}
template <class V>
__global__ __launch_bounds__(256) void cp(const V* __restrict__ src, V* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
// the bf16 storage pattern of the train kernels: 8 bytes per lane holding four bf16, decoded to four floats
__global__ __launch_bounds__(256) void rd_bf16x4(const u32x2* __restrict__ src, long n, float* __restrict__ out) {
    float s = 0.f;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += 4 * stride) {
        u32x2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + (i + u * stride < n ? i + u * stride : i));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n)
                s += __uint_as_float(v[u].x << 16) + __uint_as_float(v[u].x & 0xffff0000u) + __uint_as_float(v[u].y << 16) +
                     __uint_as_float(v[u].y & 0xffff0000u);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    const long bytes = 1L << 30;
    float *a, *b, *out;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&out, 4 * 2048 * sizeof(float)));
    CK(hipMemset(a, 0, bytes));
    CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 2048;
    auto time = [&](const char* name, auto launch, double moved) {
        launch();                                   // warm-up (not under the counters' eye when reps == 1: see below)
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-26s %8.1f GB/s  (%.3f ms per launch, %.3f GB moved)\n", name, moved * reps / (ms * 1e-3) / 1e9, ms / reps, moved / 1e9);
    };
    time("rd 4 B/lane plain", [&] { hipLaunchKernelGGL((rd<float, false>), dim3(grid), dim3(256), 0, 0, a, bytes / 4, out); }, (double)bytes);
    time("rd 4 B/lane nontemporal", [&] { hipLaunchKernelGGL((rd<float, true>), dim3(grid), dim3(256), 0, 0, a, bytes / 4, out); }, (double)bytes);
    time("rd 8 B/lane plain", [&] { hipLaunchKernelGGL((rd<f32x2, false>), dim3(grid), dim3(256), 0, 0, (const f32x2*)a, bytes / 8, out); }, (double)bytes);
    time("rd 8 B/lane nontemporal", [&] { hipLaunchKernelGGL((rd<f32x2, true>), dim3(grid), dim3(256), 0, 0, (const f32x2*)a, bytes / 8, out); }, (double)bytes);
    time("rd 16 B/lane plain", [&] { hipLaunchKernelGGL((rd<f32x4, false>), dim3(grid), dim3(256), 0, 0, (const f32x4*)a, bytes / 16, out); }, (double)bytes);
    time("rd 16 B/lane nontemporal", [&] { hipLaunchKernelGGL((rd<f32x4, true>), dim3(grid), dim3(256), 0, 0, (const f32x4*)a, bytes / 16, out); }, (double)bytes);
    time("rd 4 x bf16 (8 B/lane, nt)", [&] { hipLaunchKernelGGL(rd_bf16x4, dim3(grid), dim3(256), 0, 0, (const u32x2*)a, bytes / 8, out); }, (double)bytes);
    time("cp 4 B/lane", [&] { hipLaunchKernelGGL((cp<float>), dim3(grid), dim3(256), 0, 0, a, b, bytes / 4); }, 2.0 * bytes);
    time("cp 8 B/lane", [&] { hipLaunchKernelGGL((cp<f32x2>), dim3(grid), dim3(256), 0, 0, (const f32x2*)a, (f32x2*)b, bytes / 8); }, 2.0 * bytes);
    time("cp 16 B/lane", [&] { hipLaunchKernelGGL((cp<f32x4>), dim3(grid), dim3(256), 0, 0, (const f32x4*)a, (f32x4*)b, bytes / 16); }, 2.0 * bytes);
    return 0;
}
