// kernels.h -- launch interface between api.cpp (host) and kernels.hip (gfx950 device code).
#pragma once
#include <hip/hip_runtime.h>

namespace gtk {

// workgroup geometry: 11 waves per utterance/stream, time chunks of 16 frames.
// 16 frames x 33 bins = 528 positions = exactly 33 MFMA tiles of 16 positions = 11 waves x 3 tiles,
// so every wave owns the same number of tiles and no MFMA sits behind a branch.
constexpr int TC = 16;
constexpr int NW = 11;
constexpr int NTHR = NW * 64;
constexpr int NT2 = TC * 33 / 16;
constexpr int TPW = NT2 / NW;
// calls of at most SHORT_T frames (streaming steps) run the model kernels with ONE tile per wave: the first 11
// tiles (176 positions) cover every valid position of up to 5 frames, so two thirds of the tile work is skipped
constexpr int SHORT_T = (NW * 16) / 33;           // 5 frames: one tile per wave
constexpr int SHORT_T2 = (2 * NW * 16) / 33;      // 10 frames: two tiles per wave
static_assert(TC * 33 % 16 == 0, "chunk must be a whole number of tiles");
static_assert(NT2 % NW == 0, "tiles must divide evenly over the waves");

// STFT / iSTFT geometry
constexpr int FRAMES_PER_WAVE = 2;   // frames each of the 4 waves of a k_stft workgroup transforms
constexpr int ISTFT_BLOCKS = 7;      // hop blocks one k_istft workgroup emits (from 8 frames: 38 KB of LDS, 4 workgroups per CU)

// per-stream state (floats).  Rings are indexed by absolute frame number: the conv/TRA rings
// hold 2 rows (row = frame & 1), TCN block k holds 2d rows (row = frame mod 2d, d = 2^k).
constexpr int ST_POS = 0;                        // int frame counter (+3 pad)
constexpr int ST_ENC_H = 4;                      // [3 blocks][2][33][16]
constexpr int ST_ENC_E = ST_ENC_H + 3 * 2 * 33 * 16;   // [3][2][8]
constexpr int ST_G1_H = ST_ENC_E + 48;           // [30 rows][33][16], block k at row 2*(2^k - 1)
constexpr int ST_G2_H = ST_G1_H + 30 * 33 * 16;
constexpr int ST_DEC_H = ST_G2_H + 30 * 33 * 16;
constexpr int ST_DEC_E = ST_DEC_H + 3 * 2 * 33 * 16;
constexpr int ST_FLOATS = ST_DEC_E + 48;         // 38116 floats = 152 464 B per stream

// int8-weight / fp16-activation variant (BASELINE configs[4]); steps > 0 add the tflite path's int8 boundary
struct Quant {
    float in_step = 0.f;    // input quantiser step (calib scale / 255), 0 = fp16 input
    float out_step = 0.f;   // output quantiser step, 0 = fp16 output
};

int configure_kernels();
// lens (optional, device, int32[B]): variable-length batch -- utterance b holds lens[b] <= L samples in its row of
// L, i.e. 1 + lens[b]/256 <= T frames; L and T stay the row strides of every tensor.  nullptr: all rows are full.
int launch_stft(const float* wave, int B, long L, int T, const int* lens, const float* win, const float* twid,
                float* spec, long sb, long sf, long st, float* frames, hipStream_t s);
// 16-bit PCM <-> float32 at the host boundary (n samples, a multiple of 8; dir 0: int16 / 32768, 1: clip(rint(y * 32768)))
int launch_pcm16_convert(const void* src, void* dst, long n, int dir, hipStream_t s);
int launch_istft(const float* spec, long sb, long sf, long st, int B, int T, const int* lens, const float* win,
                 const float* twid, float* wave, hipStream_t s);
// gspec += adjoint(iSTFT)(gwave): gwave (B, 256 (T-1)) is the gradient w.r.t. the iSTFT output ALREADY divided by
// the window envelope; gspec (B,257,T,2 by strides) receives the gradient w.r.t. the spectrogram (accumulated).
int launch_istft_adjoint(const float* gwave, int B, int T, const float* win, const float* twid, float* gspec, long sb,
                         long sf, long st, hipStream_t s);
int launch_encoder(const float* spec, long sb, long sf, long st, int B, int T, const int* lens, const float* PF,
                   const int* PI, float* en0, float* en1, float* en2, float* en3, float* en4, float* state,
                   unsigned long long* stamps, hipStream_t s, const Quant* q = nullptr, bool front_done = false,
                   const int* pref = nullptr);
// offline front end: STFT (wave) or caller spectrogram -> features, ERB, SFE, en_conv0, en_conv1 (see kernels.hip)
int launch_front(const float* wave, long L, const float* spec_in, long isb, long isf, long ist, int B, int T,
                 const int* lens, const float* win, const float* twid, const float* PF, const int* PI, float* spec_out,
                 float* en0, float* en1p, hipStream_t s, const Quant* q = nullptr);
int launch_gtcn(const float* xin, float* xout, const float* P, int B, int T, float* state, int st_off,
                const float* addend, unsigned long long* stamps, hipStream_t s);
int launch_gtcn_ms(const float* xin, float* xout1, float* xout2, const float* P, int B, float* state, hipStream_t s);
int launch_gtcn_band(const float* xin, float* xout, const float* P, int B, int T, const int* lens, const float* addend,
                     hipStream_t s, const Quant* q = nullptr, const int* pref = nullptr);
int launch_decoder(const float* xg, const float* en0, const float* en1, const float* en2, const float* en3,
                   const float* en4, const float* spec, long sb, long sf, long st, float* out, long osb, long osf,
                   long ost, int B, int T, const int* lens, const float* PF, const int* PI, float* state, float* dbg,
                   unsigned long long* stamps, hipStream_t s, const Quant* q = nullptr, const int* pref = nullptr);
// Variable-length batches in time spans: pref (device, int32[B + 1]) = prefix sums of the utterances' frame counts, made
// from lens by launch_len_prefix (B <= 1024: var_spans_usable); the three per-utterance launchers above then run whole
// rounds of 256 workgroups, each taking an equal share of the frames that exist, instead of one workgroup per utterance.
bool var_spans_usable(int B);
int launch_len_prefix(const int* lens, int B, int T, int* pref, hipStream_t s);
// single-frame streaming step of B streams as ONE launch (encoder -> both GTCN stacks -> decoder, nothing through HBM)
int launch_stream_ms(const float* spec, long sb, long sf, float* out, long osb, long osf, int B, const float* PF,
                     const int* PI, float* state, unsigned long long* stamps, hipStream_t s);
// the same step with stream_wide_streams() streams per workgroup (eight waves x two tiles, streamed parameters): the form
// for stream counts that fill the chip more than once
int launch_stream_wide(const float* spec, long sb, long sf, float* out, long osb, long osf, int B, const float* PF,
                       const int* PI, float* state, unsigned long long* stamps, hipStream_t s);
int stream_wide_streams();
bool stream_ms_usable(long sb, long osb);
int launch_state_convert(float* state, int N, float* conv, float* tra, float* const* tcn8, const int* PI, int dir,
                         hipStream_t s);
int launch_conv2d_causal(const float* x, const float* cache, const float* w, const float* bias, float* y,
                         float* cache_out, int B, int Cin, int Cout, int T, int F, int kt, int kf, int dt, int df,
                         int pf, int groups, int transposed, int Fout, hipStream_t s);
int launch_selftest(const float* A, const float* Bm, const float* C, float* D, hipStream_t s);
// n (a multiple of 4) values -> planes [3][n] and joined [n]; A (16x32), Bm (32x16) -> D (16x16) through split_mm6
int launch_selftest_split3(const float* x, long n, float* planes, float* joined, const float* A, const float* Bm,
                           float* D, hipStream_t s);

}  // namespace gtk
