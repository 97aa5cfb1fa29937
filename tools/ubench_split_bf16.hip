// Microbenchmark behind DESIGN.md section 4 ("dense 3x3 on the 16-bit matrix pipe"): the decoder's dense transposed
// 3x3 (a 16 x 144 by 144 x 16 product per 16-position tile) as
//   F32   36 x v_mfma_f32_16x16x4_f32 (what the kernels did up to round 2), or
//   S3    a three-way bf16 split of both operands (x = hi + mid + lo EXACTLY: 3 x 8 significant bits = fp32's 24) and
//         the SIX products whose weight is >= 2^-16 (hh, hm, mh, mm, hl, lh) on v_mfma_f32_16x16x32_bf16: 5 K-chunks of
//         two taps x 6 = 30 instructions of 16 cycles that co-execute with the VALU, or
//   S2    a two-way split with three products (hh, hl, lh): 15 instructions, 2^-17 class error (for reference only), or
//   WINO  Winograd F(2,3) along frequency in fp32 (the other route VERDICT r2 asked to be measured): a twin tile = 16
//         PAIRS of adjacent bins; per kernel row four transformed inputs (d0-d2, d1+d2, d2-d1, d1-d3: 16 vector ops)
//         times four transformed 16x16 weight matrices (16 fp32 MFMAs instead of 24), one output transform per twin
//         (16 vector ops): 48 fp32 MFMAs + 64 vector ops per 32 positions instead of 72 MFMAs.
// Part 1 (numerics): one wave, random operands, error of each form against the float64 product.
// Part 2 (timing): the dense phase as the decoder runs it -- 11 waves, three tiles per wave, taps and weights read
//   from LDS with ds_read_b128 (96-byte records either way), optional VALU filler per tile -- one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_split_bf16.hip -o /tmp/ubs && /tmp/ubs
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf_hi(float x) { return (float)(__bf16)x; }   // RNE to 8 significant bits
__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;      // exact
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;     // exact, <= 8 significant bits
    l = (__bf16)r2;
}

// ---------------------------------------------------------------------------------------------- part 1: numerics
// W [16][144] (row m, k), H [144][16] (k, column n); D [mode][16][16].  k = tap * 16 + channel.
__global__ void k_num(const float* __restrict__ W, const float* __restrict__ H, float* __restrict__ D) {
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    // F32: lane (n, g): a = A[row n][k = g], b = B[k = g][col n]
    f32x4 acc = {0, 0, 0, 0};
    for (int kk = 0; kk < 36; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[n * 144 + 4 * kk + g], H[(4 * kk + g) * 16 + n], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[0 * 256 + (4 * g + r) * 16 + n] = acc[r];
    // split forms: lane (n, g) holds k = 8g .. 8g+7 of row n (A) / column n (B)
    f32x4 s3 = {0, 0, 0, 0}, s3big = {0, 0, 0, 0}, s3small = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, s9 = {0, 0, 0, 0};
    for (int c = 0; c < 5; ++c) {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * c + 8 * g + j;
            const float a = k < 144 ? W[n * 144 + k] : 0.f, b = k < 144 ? H[k * 16 + n] : 0.f;
            __bf16 h, m, l;
            split3(a, h, m, l); ah[j] = h; am[j] = m; al[j] = l;
            split3(b, h, m, l); bh[j] = h; bm[j] = m; bl[j] = l;
        }
#define MM(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, C_, 0, 0, 0)
        MM(ah, bh, s3); MM(ah, bm, s3); MM(am, bh, s3); MM(am, bm, s3); MM(ah, bl, s3); MM(al, bh, s3);
        MM(ah, bh, s3big);
        MM(ah, bm, s3small); MM(am, bh, s3small); MM(am, bm, s3small); MM(ah, bl, s3small); MM(al, bh, s3small);
        MM(ah, bh, s9); MM(ah, bm, s9); MM(am, bh, s9); MM(am, bm, s9); MM(ah, bl, s9); MM(al, bh, s9);
        MM(am, bl, s9); MM(al, bm, s9); MM(al, bl, s9);
        // two-way split: lo2 = bf16(x - hi)
        bf16x8 al2, bl2;
        for (int j = 0; j < 8; ++j) { al2[j] = (__bf16)((float)am[j] + (float)al[j]); bl2[j] = (__bf16)((float)bm[j] + (float)bl[j]); }
        MM(ah, bh, s2); MM(ah, bl2, s2); MM(al2, bh, s2);
#undef MM
    }
    for (int r = 0; r < 4; ++r) {
        D[1 * 256 + (4 * g + r) * 16 + n] = s3[r];
        D[2 * 256 + (4 * g + r) * 16 + n] = s3big[r] + s3small[r];
        D[3 * 256 + (4 * g + r) * 16 + n] = s2[r];
        D[4 * 256 + (4 * g + r) * 16 + n] = s9[r];
    }
}

// ---------------------------------------------------------------------------------------------- part 2: timing
constexpr int NW = 11, NTHR = NW * 64, RS = 24, PT = 35, ROWS = 18;
constexpr int IMG = ROWS * PT * RS;                  // floats
constexpr int AF32 = 9 * 256;                        // floats
constexpr int ABF = 5 * 3 * 16 * 32 / 2;             // floats (bf16 pairs)
// MODE 0: fp32; 1: split-3 (6 products); 2: split-2 (3 products).  FILL: extra independent v_fma per tile and tap/chunk.
template <int MODE, int FILL>
__global__ __launch_bounds__(NTHR) void k_time(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sW = sm;
    float* sA = sm + IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    for (int i = tid; i < IMG + AF32 + ABF; i += NTHR) sm[i] = 0.f;   // zeros: finite in either interpretation
    __syncthreads();
    int b0[3];
    for (int i = 0; i < 3; ++i) {
        const int p = (wave + i * NW) * 16 + n, tl = p / 33, ff = p - tl * 33;
        b0[i] = ((tl + 2) * PT + 1 + ff) * RS;
    }
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float fill[8];
    for (int q = 0; q < 8; ++q) fill[q] = q + lane;
    const float fa = 1.0f + lane * 1e-7f, fb = 1e-3f;
    for (int it = 0; it < iters; ++it) {
        int o0 = b0[0], o1 = b0[1], o2 = b0[2];
        asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2));
        const int oo[3] = {o0, o1, o2};
        if (MODE == 0) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kt = tap / 3, kf = tap % 3;
                const f32x4 A = *reinterpret_cast<const f32x4*>(sA + tap * 256 + n * 16 + 4 * (g ^ ((n >> 1) & 2)));
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(sW + oo[i] - kt * PT * RS + (1 - kf) * RS + 4 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s], t[s], acc[i], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < FILL; ++q) fill[q & 7] = fmaf(fill[q & 7], fa, fb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 3) {
            // Winograd F(2,3): the wave's three accumulator sets are three TWIN tiles (= six plain tiles per iteration);
            // lane n holds the pair of bins (2n, 2n+1): records at a stride of two
            f32x4 m[3][4];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) m[i][j] = acc[i];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                f32x4 A[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) A[j] = *reinterpret_cast<const f32x4*>(sA + ((kt * 4 + j) % 9) * 256 + n * 16 + 4 * (g ^ ((n >> 1) & 2)));
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float* base = sW + (1 + (oo[i] / RS + n + kt * PT) % (ROWS * PT - 4)) * RS + 4 * g;   // the pair (2n, 2n+1): record stride two
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(base - RS), d1 = *reinterpret_cast<const f32x4*>(base),
                                d2 = *reinterpret_cast<const f32x4*>(base + RS), d3 = *reinterpret_cast<const f32x4*>(base + 2 * RS);
                    const f32x4 u[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int s = 0; s < 4; ++s) m[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j][s], u[j][s], m[i][j], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < FILL * 6; ++q) fill[q & 7] = fmaf(fill[q & 7], fa, fb);   // 9 FILL per tile: a twin is two tiles, three kernel rows
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] = (m[i][0] + m[i][1] + m[i][2]) + (m[i][1] - m[i][2] - m[i][3]);   // y0 + y1 (kept live)
        } else {
            constexpr int NP = MODE == 1 ? 3 : 2;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int tA = 2 * c, tB = 2 * c + 1 < 9 ? 2 * c + 1 : 8;
                const int offA = -(tA / 3) * PT * RS + (1 - tA % 3) * RS, offB = -(tB / 3) * PT * RS + (1 - tB % 3) * RS;
                const int off = (g >= 2 ? offB : offA) + 4 * (g & 1);
                bf16x8 ap[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    ap[p] = *reinterpret_cast<const bf16x8*>(sA + AF32 + ((c * 3 + p) * 16 + n) * 16 + 4 * g);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    bf16x8 bp[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) bp[p] = *reinterpret_cast<const bf16x8*>(sW + oo[i] + off + 8 * p);
#define MM(A_, B_) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, acc[i], 0, 0, 0)
                    if (MODE == 1) { MM(ap[0], bp[2]); MM(ap[2], bp[0]); MM(ap[1], bp[1]); MM(ap[0], bp[1]); MM(ap[1], bp[0]); MM(ap[0], bp[0]); }
                    else { MM(ap[0], bp[1]); MM(ap[1], bp[0]); MM(ap[0], bp[0]); }
#undef MM
#pragma unroll
                    for (int q = 0; q < FILL * 9 / 5; ++q) fill[q & 7] = fmaf(fill[q & 7], fa, fb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int q = 0; q < 8; ++q) s += fill[q];
    out[blockIdx.x * NTHR + tid] = s;
}

template <int MODE, int FILL>
float run(float* d, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int lds = (IMG + AF32 + ABF) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_time<MODE, FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k_time<MODE, FILL>), dim3(256), dim3(NTHR), lds, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k_time<MODE, FILL>), dim3(256), dim3(NTHR), lds, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    // ---- numerics
    const int NTRIAL = 64;
    float *dW, *dH, *dD;
    hipMalloc(&dW, 16 * 144 * 4); hipMalloc(&dH, 144 * 16 * 4); hipMalloc(&dD, 5 * 256 * 4);
    std::vector<float> W(16 * 144), H(144 * 16), D(5 * 256);
    double emax[5] = {0, 0, 0, 0, 0}, erms[5] = {0, 0, 0, 0, 0};
    srand(1);
    auto rnd = [] { double u = 0; for (int i = 0; i < 12; ++i) u += rand() / (double)RAND_MAX; return u - 6.0; };
    for (int tr = 0; tr < NTRIAL; ++tr) {
        for (auto& w : W) w = (float)(rnd() * 0.1);
        for (auto& h : H) h = (float)(rnd() * (tr & 1 ? 1.0 : 30.0) + (tr & 2 ? 0.7 : 0.0));   // PReLU-like offsets too
        hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_num, dim3(1), dim3(64), 0, 0, dW, dH, dD);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        double ref[256], scale = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double s = 0;
                for (int k = 0; k < 144; ++k) s += (double)W[m * 144 + k] * (double)H[k * 16 + n];
                ref[m * 16 + n] = s;
                scale = fmax(scale, fabs(s));
            }
        for (int mode = 0; mode < 5; ++mode)
            for (int i = 0; i < 256; ++i) {
                const double e = fabs(D[mode * 256 + i] - ref[i]) / scale;
                emax[mode] = fmax(emax[mode], e);
                erms[mode] += e * e / (256.0 * NTRIAL);
            }
    }
    const char* names[5] = {"fp32 16x16x4 chain (36)", "bf16 split-3, 6 products, one accumulator (30)",
                            "bf16 split-3, 6 products, big + small accumulators (30)", "bf16 split-2, 3 products (15)",
                            "bf16 split-3, all 9 products (45)"};
    printf("numerics vs float64 (error / max|D| over %d random 16x144x16 products):\n", NTRIAL);
    for (int mode = 0; mode < 5; ++mode) printf("  %-58s max %.3e  rms %.3e\n", names[mode], emax[mode], sqrt(erms[mode]));
    // ---- timing
    float* d; hipMalloc(&d, 256 * NTHR * 4);
    const int iters = 4000;
    // per iteration: 33 tiles per CU; per SIMD (3/3/3/2 waves): up to 9 tiles
    auto rep = [&](const char* name, float ms, int mfma_per_tile, int cyc) {
        const double cyc_tile_simd = ms * 1e-3 * 2.4e9 / (iters * 9.0);
        printf("  %-44s %.3f ms  -> %.0f cycles per tile on the 3-wave SIMDs @2.4 GHz (matrix-pipe floor %d)\n", name, ms,
               cyc_tile_simd, mfma_per_tile * cyc);
    };
    printf("timing (256 workgroups x 11 waves x 3 tiles, %d iterations):\n", iters);
    rep("fp32, no filler", run<0, 0>(d, iters), 36, 32);
    rep("fp32, 3 v_fma per tile and tap (27/tile)", run<0, 3>(d, iters), 36, 32);
    rep("fp32, 9 v_fma per tile and tap (81/tile)", run<0, 9>(d, iters), 36, 32);
    rep("split-3 x6, no filler", run<1, 0>(d, iters), 30, 16);
    rep("split-3 x6, 27 v_fma per tile", run<1, 3>(d, iters), 30, 16);
    rep("split-3 x6, 81 v_fma per tile", run<1, 9>(d, iters), 30, 16);
    rep("split-2 x3, no filler", run<2, 0>(d, iters), 15, 16);
    rep("split-2 x3, 81 v_fma per tile", run<2, 9>(d, iters), 15, 16);
    // Winograd: an iteration covers SIX tile equivalents per wave (three twins): per-tile figures = measured / 2
    {
        const float w0 = run<3, 0>(d, iters), w9 = run<3, 9>(d, iters);
        printf("  %-44s %.3f ms  -> %.0f cycles per TILE EQUIVALENT (24 fp32 MFMAs + 32 transform ops; matrix-pipe floor 768)\n",
               "Winograd F(2,3) fp32, no filler", w0, w0 * 1e-3 * 2.4e9 / (iters * 18.0));
        printf("  %-44s %.3f ms  -> %.0f cycles per TILE EQUIVALENT\n", "Winograd F(2,3) fp32, 81 v_fma per tile", w9,
               w9 * 1e-3 * 2.4e9 / (iters * 18.0));
    }
    // ---- Winograd numerics (host, fp32 arithmetic in the MFMA chain order: channels inside a kernel row) vs float64
    {
        double emax_d = 0, emax_w = 0, erms_d = 0, erms_w = 0;
        const int P = 32, NTR = 64;
        std::vector<float> Wt(9 * 16 * 16), Hh(3 * (P + 2) * 16);
        for (int tr = 0; tr < NTR; ++tr) {
            for (auto& w : Wt) w = (float)(rnd() * 0.1);
            for (auto& h : Hh) h = (float)(rnd() * (tr & 1 ? 1.0 : 30.0) + (tr & 2 ? 0.7 : 0.0));
            auto Hat = [&](int kt, int p, int i) { return Hh[(kt * (P + 2) + p + 1) * 16 + i]; };   // p in [-1, P]
            std::vector<double> ref(P * 16);
            std::vector<float> dir(P * 16), win(P * 16);
            double scale = 0;
            for (int p = 0; p < P; ++p)
                for (int o = 0; o < 16; ++o) {
                    double s = 0; float f = 0.f;
                    for (int kt = 0; kt < 3; ++kt)
                        for (int kf = 0; kf < 3; ++kf)
                            for (int i = 0; i < 16; ++i) {
                                const float w = Wt[((kt * 3 + kf) * 16 + o) * 16 + i], h = Hat(kt, p + 1 - kf, i);
                                s += (double)w * h;
                                f = fmaf(w, h, f);
                            }
                    ref[p * 16 + o] = s; dir[p * 16 + o] = f; scale = fmax(scale, fabs(s));
                }
            for (int p = 0; p < P; p += 2)
                for (int o = 0; o < 16; ++o) {
                    float m[4] = {0, 0, 0, 0};
                    for (int kt = 0; kt < 3; ++kt)
                        for (int i = 0; i < 16; ++i) {
                            // y[p] = g0 d[p+1] + g1 d[p] + g2 d[p-1] with g_kf = W[kt][kf]: as a correlation over
                            // e0..e3 = d[p-1], d[p], d[p+1], d[p+2] the taps are (g2, g1, g0)
                            const float g0 = Wt[((kt * 3 + 2) * 16 + o) * 16 + i], g1 = Wt[((kt * 3 + 1) * 16 + o) * 16 + i],
                                        g2 = Wt[((kt * 3 + 0) * 16 + o) * 16 + i];
                            const float e0 = Hat(kt, p - 1, i), e1 = Hat(kt, p, i), e2 = Hat(kt, p + 1, i), e3 = Hat(kt, p + 2, i);
                            const float G[4] = {g0, 0.5f * (g0 + g1 + g2), 0.5f * (g0 - g1 + g2), g2};
                            const float U[4] = {e0 - e2, e1 + e2, e2 - e1, e1 - e3};
                            for (int j = 0; j < 4; ++j) m[j] = fmaf(G[j], U[j], m[j]);
                        }
                    win[p * 16 + o] = m[0] + m[1] + m[2];
                    win[(p + 1) * 16 + o] = m[1] - m[2] - m[3];
                }
            for (int k = 0; k < P * 16; ++k) {
                const double ed = fabs(dir[k] - ref[k]) / scale, ew = fabs(win[k] - ref[k]) / scale;
                emax_d = fmax(emax_d, ed); emax_w = fmax(emax_w, ew);
                erms_d += ed * ed / (P * 16.0 * NTR); erms_w += ew * ew / (P * 16.0 * NTR);
            }
        }
        printf("Winograd F(2,3) numerics (host fp32 emulation, %d trials): direct fp32 chain max %.3e rms %.3e | Winograd max %.3e rms %.3e\n",
               NTR, emax_d, sqrt(erms_d), emax_w, sqrt(erms_w));
    }
    return 0;
}
