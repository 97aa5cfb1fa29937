// train_kernels.h -- launch interface between train.cpp (host) and train_kernels.hip (gfx950):
// the train-mode forward/backward of the GTCRN-Micro model (reference: models/gtcrn_micro.py with
// nn.BatchNorm2d in .train() mode, train.py:239-288).
//
// Unlike the inference kernels (kernels.hip), training is layer-at-a-time: train-mode BatchNorm
// needs the statistics of a layer's output over the WHOLE batch before the next layer can run,
// so every conv -> BN -> activation unit is a grid-wide pass over channels-last tensors
// [B][T][F][C] (C = 2, 3, 8 or 16 floats per position), and the passes are HBM-streaming.
#pragma once
#include <hip/hip_runtime.h>

namespace gtt {

// A convolution over (time, frequency) on channels-last tensors, gather form:
//   out[b, to, fo, cout_off + co] (+)= bias[co] + sum_{kt,kf,ci} W[co,ci,kt,kf] * in[b, to + t_off[kt], fi, cin_off + ci]
//   f_mode 0 (Conv2d along F):           fi = fo * sf - pf + kf
//   f_mode 1 (ConvTranspose2d along F):  fi = (fo + pf - kf) / sf   when divisible
// Out-of-range taps read zero.  The adjoint (data gradient) of such a convolution is again one of
// these with in/out swapped, t_off negated, f_mode flipped and the weight strides of ci/co swapped,
// so one kernel serves Conv2d, ConvTranspose2d and both their data gradients.
struct ConvGeom {
    int B, Tin, Tout, Fin, Fout;
    int CinT, cin_off, Cin;     // channels of the input tensor, first channel used, channels used
    int CoutT, cout_off, Cout;
    int nkt, nkf;
    int t_off[3];
    int f_mode, sf, pf;
    int w_co, w_ci, w_kt, w_kf; // strides (floats) into the weight tensor in the reference's layout
    int accumulate;             // 1: add to the existing output instead of overwriting it
    int in_bf, out_bf;          // storage of `in` / `out`: 0 fp32, 1 bf16 (saved activations of the bf16 variant);
                                // gradients are always fp32; out_bf excludes accumulate
    float* out2;                // (unused by the dense convs: see DwGeom)
    int out2_bf;
};
// Storage mode "bf16 saves, exact forward chain" (gtcrn_trainer_set_storage 4): the forward runs on fp32 tensors exactly as
// in mode 0 (it IS the fp32 network, bit for bit), and every tensor the backward re-reads is ALSO written as a 16-bit
// copy: activations / block outputs / features by their producer (a second store), conv outputs y by their CONSUMER
// (bn_act or the normalise-on-load conv), which knows this step's statistics and stores bf16(y - mean) on the same side
// of the PReLU kink as the forward's value (ycopy_value in train_kernels.hip).  Only the forward kernels know about the
// mode; the backward is the bf16 mode's, with statistics (0, invstd) for the centred copies.

// gbf (the backward entry points): storage format of the GRADIENT tensors handed between units (da / dx / dres / dout /
// dv): 0 fp32 (every mode up to round 4), 1 bf16 (storage mode 5, with bf = ybf = 1): a unit rounds the gradient it hands
// on to bf16 where it stores it -- what autocast-style bf16 training does -- which halves the remaining fp32 streams of
// the bf16 mode; the arithmetic inside a unit, the BatchNorm reductions and every parameter gradient stay fp32.
// Storage of the SAVED tensors (conv outputs, activations, block outputs -- everything the backward re-reads):
// the format arguments of the functions below (`bf`: activations / block outputs, `ybf`: conv outputs in front of a
// BatchNorm; in_bf / out_bf of the geometries) are 0 = fp32 (the reference's precision), 1 = bf16 (the saturating fp16
// code 2 of the diagnostic storage modes of rounds 2-4 is gone: its run-time format switch made every 16-bit conversion
// compute both encodings).  A 16-bit tensor is passed through the same float* handle; its elements are 2 bytes wide.
// Arithmetic, statistics, gradients and weights are fp32 throughout.

// depthwise convolution (groups = channels), C channels:
//   out[b,to,fo,c] (+)= bias[c] + sum_{kt,kf} W[c,kt,kf] * in[b, to + t_off[kt], fo + f_off[kf], c]
struct DwGeom {
    int B, Tin, Tout, F, C;
    int nkt, nkf;
    int t_off[3], f_off[3];
    int w_c, w_kt, w_kf;
    int accumulate;
    int in_bf, out_bf;
    float* out2;                // exact chain: second copy of the output in format out2_bf (`out` is fp32 then); the
    int out2_bf;                // 3-channel SFE conv only (its output is a conv input the backward re-reads, no BatchNorm)
};

enum Act { ACT_NONE = 0, ACT_PRELU = 1, ACT_TANH = 2 };

constexpr int MAX_PARTIALS = 1024;   // workgroups of a two-stage reduction

// In-launch finish of the BatchNorm reductions (fusion bit 10): the last workgroup of the kernel that produces the
// per-workgroup sums also adds them up and evaluates the result (statistics + running estimates in the forward; the
// means, dgamma, dbeta, dslope in the backward) -- no second-stage kernel (92 of them per step before).  The trainer
// owns the buffers: FIN_GPART_DOUBLES doubles of group sums and FIN_CTR_WORDS counters that are ZERO between launches
// (allocated zeroed, reset by the last arrivers, zeroed again at the start of every forward / backward call).
constexpr int FIN_GPART_DOUBLES = 32 * 64;
constexpr int FIN_CTR_WORDS = 64;
void set_fin_context(bool on, double* gpart, unsigned* ctr);      // thread local; on = false: finish kernels as before
// fusion bit 15: the fused backward launchers record their weight-gradient finishes (partials in a region of `pool`) instead of
// launching them; flush_wgrad_finishes runs them in batches (thread local; on = false or pool exhausted: as before)
void set_wgrad_defer(bool on, float* pool, size_t cap_floats);
int flush_wgrad_finishes(hipStream_t s);
void set_column_form(bool on);        // thread local; fusion bit 14: dw_fwd sends the TCN's dilated (3,1) conv through k_dw31_col
void set_pointwise_form(bool on);     // thread local; fusion bit 13: conv_fwd sends pure 1x1 convs through k_pw_fwd
// what the statistics finish of a forward unit needs; given to conv_fwd / dw_fwd (sf), they run it in the producing
// kernel when the context is on: *stat_parts is then NEGATIVE and bn_stats(have_parts < 0) has nothing left to do
struct StatFin {
    long n;
    int C;
    float *stats, *running_mean, *running_var, *shift, *stats_b;
};

// stat_partial/stat_parts (optional): the kernel also leaves per-workgroup sums of out and out^2 per channel
// (layout of bn_stats' scratch) and reports their count, so the BatchNorm statistics need no extra pass.
// shift (optional, bf16 output only): out = conv(in) + bias - shift[c]; the following train-mode BatchNorm is shift
// invariant.  bn_stats(..., shift) accounts for it in the running-mean update and sets shift[c] to this step's batch
// mean for the next step, so the stored values stay centred (bf16 rounding relative to the spread, not the offset).
// pre (optional; 1x1 convs over all 16 channels and the 16-channel depthwise convs with (3,1) taps): `in` is the
// PREVIOUS unit's conv output y (format pre->ybf); the kernel applies that unit's BatchNorm + PReLU while loading
// (same expressions as bn_act: bit-identical values), writes the activation to pre->a_out (format pre->bf) -- the
// backward reads it -- and convolves the stored (rounded) activation.  The separate bn_act pass of the previous unit
// (read y, write a) disappears; its consumer reads y instead of a.
struct BnPre {
    const float *stats, *gamma, *beta, *slope;   // of the previous unit; slope == nullptr: no activation
    float* a_out;
    int ybf, bf;
    const float* res;                            // its residual input (format bf) or nullptr; pointwise convs only
    // exact chain: `in` is fp32, the activation enters the convolution UNROUNDED, a_out still receives its 16-bit copy,
    // a_chain (optional) the fp32 value for the other readers of this activation (residual, skip), res is fp32; y_out
    // (format ybf_out) receives the centred 16-bit copy of the previous unit's conv output for the backward
    int exact;
    float* a_chain;
    float* y_out;
    int ybf_out;
};
struct DwUnitNext;
// next (backward launches only): the output is the gradient input of that unit -- its BatchNorm reduction rides in the
// epilogue (see DwUnitNext below; 3x3 and 1x5 dense adjoints, the depthwise 3x3 adjoint), per-workgroup sums in
// stat_partial, their count in *stat_parts; next_yfmt: the storage format of that unit's y
int conv_fwd(const ConvGeom& g, const float* in, const float* w, const float* bias, float* out, hipStream_t s,
             double* stat_partial = nullptr, int* stat_parts = nullptr, const float* shift = nullptr,
             const BnPre* pre = nullptr, const DwUnitNext* next = nullptr, int next_yfmt = 0,
             const StatFin* sf = nullptr);
// dW (and db when dbias != nullptr) of the convolution g: in = its input, dout = gradient of its output.
// scratch: MAX_PARTIALS * (9*256 + 16) floats.
int conv_wgrad(const ConvGeom& g, const float* in, const float* dout, float* dw, float* dbias, float* scratch,
               hipStream_t s);
int dw_fwd(const DwGeom& g, const float* in, const float* w, const float* bias, float* out, hipStream_t s,
           double* stat_partial = nullptr, int* stat_parts = nullptr, const float* shift = nullptr,
           const BnPre* pre = nullptr, const DwUnitNext* next = nullptr, int next_yfmt = 0,
           const StatFin* sf = nullptr);
int dw_wgrad(const DwGeom& g, const float* in, const float* dout, float* dw, float* dbias, float* scratch,
             hipStream_t s);

// BatchNorm (train mode) of y [n][C]: batch statistics, running-statistics update (momentum 0.1, unbiased
// variance), stats[0..C) = mean, stats[C..2C) = 1/sqrt(var + 1e-5).  scratch: MAX_PARTIALS * 2 * C doubles.
// stats_b (exact chain only): the statistics the BACKWARD uses for the 16-bit copy of y, which its consumer writes as
// bf16(y - mean): (0, invstd).
int bn_stats(const float* y, long n, int C, float* stats, float* running_mean, float* running_var, double* scratch,
             hipStream_t s, int have_parts = 0, int bf = 0, float* shift = nullptr, float* stats_b = nullptr);
// a = act(gamma * (y - mean) * invstd + beta [+ res]); exact chain: a2 (format a2_bf) = second copy of a, y2 (format
// y2_bf) = the centred copy of y for the backward.  post (format bf; not with a2 / y2): added AFTER the activation -- what is
// stored is a + post, the next decoder layer's input x + skip (fusion bit 11: the activation has no other reader)
int bn_act(const float* y, long n, int C, const float* stats, const float* gamma, const float* beta,
           const float* res, int act, const float* slope, float* a, hipStream_t s, int bf = 0, int ybf = 0,
           float* a2 = nullptr, int a2_bf = 0, float* y2 = nullptr, int y2_bf = 0, const float* post = nullptr);
// backward of the same: given da, writes dy (may alias da); if dres != nullptr: dres (+)= dz (dres_acc: add);
// dgamma/dbeta [C], dslope [1] (PReLU) are WRITTEN.  scratch: MAX_PARTIALS * 3 * C doubles + 2 * C floats.
int bn_act_bwd(const float* da, const float* y, long n, int C, const float* stats, const float* gamma,
               const float* beta, const float* res, int act, const float* slope, float* dy, float* dres,
               int dres_acc, float* dgamma, float* dbeta, float* dslope, double* scratch, hipStream_t s, int bf = 0,
               int ybf = 0, int have_parts = 0, int gbf = 0);

// the unit in FRONT of a unit in the backward order, whose gradient input is that unit's dx: see `next` below
struct DwUnitNext {
    const float *y, *stats, *gamma, *beta, *slope;    // of the unit in front (16 channels, PReLU)
    const float* res = nullptr;                       // its residual input (unit1x1_bwd only)
    // 1 / 2: that unit's activation IS the input x of the unit whose backward runs, and x is recomputed from y (and
    // res) instead of being read (x may then be nullptr); 2: rounded to bf16 as the forward's consumer saw it
    int recompute_x = 0;
    // in-launch finish (fusion bit 10): where that unit's parameter gradients go and its position count; with n > 0 the
    // producing kernel's last workgroup also finishes the reduction and *next_parts comes back NEGATIVE (-1 - slot of
    // the means, see parts_slot in train_kernels.hip): hand it to that unit's backward as have_parts
    float *dgamma = nullptr, *dbeta = nullptr, *dslope = nullptr;
    long n = 0;
};
// The whole backward of a pointwise (1x1) conv + BatchNorm + activation unit in two passes: the BatchNorm
// reduction, then ONE kernel that forms dy, the data gradient dx (nullptr: not needed; dx_acc: add) and the
// weight/bias gradients; dres (optional) receives dz for the residual branch.  Gradients of gamma, beta, the
// PReLU slope, the conv weight and bias are WRITTEN.
int unit1x1_bwd(const ConvGeom& g, const float* x, const float* y, const float* da, const float* res,
                const float* stats, const float* gamma, const float* beta, int act, const float* slope,
                const float* w, float* dx, int dx_acc, float* dres, int dres_acc, float* dw, float* dbias,
                float* dgamma, float* dbeta, float* dslope, double* dscratch, float* fscratch, hipStream_t s,
                int bf = 0, int ybf = 0, int have_parts = 0, const DwUnitNext* next = nullptr,
                int* next_parts = nullptr, int gbf = 0);
// The backward of a 16-channel depthwise (3,1) conv + BatchNorm + PReLU unit (TCN conv2) in two passes: the BatchNorm
// reduction, then ONE kernel for dy (never stored), the weight / bias gradient and the data gradient dx.  next
// (optional): the unit in front, whose gradient input is this dx -- its BatchNorm reduction is accumulated by the same
// kernel (per-workgroup sums left in dscratch, their count in *next_parts: hand it to unit1x1_bwd as have_parts).
int dwunit_bwd(const DwGeom& g, const float* x, const float* y, const float* da, const float* stats,
               const float* gamma, const float* beta, const float* slope, const float* w, float* dx, float* dw,
               float* dbias, float* dgamma, float* dbeta, float* dslope, double* dscratch, float* fscratch,
               hipStream_t s, int bf, int ybf, const DwUnitNext* next, int* next_parts, int have_parts = 0, int gbf = 0);

// The same for the 16-channel depthwise 3x3 unit of the encoder's GTConv blocks (F == 33; taps t-2..t, f-1..f+1): one
// LDS-tiled kernel for dy, the weight / bias gradient and dx.
int dwunit33_bwd(const DwGeom& g, const float* x, const float* y, const float* da, const float* stats,
                 const float* gamma, const float* beta, const float* slope, const float* w, float* dx, float* dw,
                 float* dbias, float* dgamma, float* dbeta, float* dslope, double* dscratch, float* fscratch,
                 hipStream_t s, int bf, int ybf, const DwUnitNext* next, int* next_parts, int have_parts = 0, int gbf = 0);

// ... and for the decoder's dense transposed 3x3 unit (ConvTranspose2d(16,16,(3,3)): 33 bins, T + 2 output frames): both
// matrix products (data gradient, weight gradient) from LDS images of dy and x.
int dense33_bwd(const ConvGeom& g, const float* x, const float* y, const float* da, const float* stats,
                const float* gamma, const float* beta, const float* slope, const float* w, float* dx, float* dw,
                float* dbias, float* dgamma, float* dbeta, float* dslope, double* dscratch, float* fscratch,
                hipStream_t s, int bf, int ybf, const DwUnitNext* next, int* next_parts, int have_parts = 0, int gbf = 0);

// ... and for the two 16 -> 16 (1,5) stride-2 units, en_convs.1 (Conv2d, 65 -> 33 bins) and de_convs.3 (ConvTranspose2d,
// 33 -> 65 bins): dy and x of eight frames in LDS, weight gradient and data gradient from there; dx may accumulate.
int conv15_bwd(const ConvGeom& g, const float* x, const float* y, const float* da, const float* stats,
               const float* gamma, const float* beta, const float* slope, const float* w, float* dx, int dx_acc,
               float* dw, float* dbias, float* dgamma, float* dbeta, float* dslope, double* dscratch, float* fscratch,
               hipStream_t s, int bf, int ybf, const DwUnitNext* next, int* next_parts, int have_parts = 0, int gbf = 0);

// features: spec (strided (b,f,t) + re/im) -> EB [B][T][129][3] = ERB.bm([mag,re,im]) (models/gtcrn_micro.py:510-516)
int feat_fwd(const float* spec, long sb, long sf, long st, int B, int T, const float* erb_w, float* eb,
             hipStream_t s, int bf = 0, float* eb2 = nullptr, int eb2_bf = 0);
// mask: m [B][T][129][2] -> ERB.bs -> complex ratio mask applied to spec (:472-482, 526-530)
int bs_mask_fwd(const float* m, const float* spec, long sb, long sf, long st, int B, int T, const float* ierb_w,
                float* out, long ob, long of, long ot, hipStream_t s, int bf = 0);
int bs_mask_bwd(const float* dout, long ob, long of, long ot, const float* spec, long sb, long sf, long st, int B,
                int T, const float* ierb_w, float* dm, hipStream_t s);

// point_bn2 on load (fusion bit 12): with tb != nullptr the `v` of the three TRALite entry points below is point_conv2's conv
// OUTPUT (format tb->ybf) and they apply point_bn2 (no activation follows it) themselves, rounding to `bf` as the stored
// activation was -- same values as the separate normalise pass, which disappears together with the stored activation
struct TraBn {
    const float *stats, *gamma, *beta;     // mean[8] invstd[8], weight, bias of point_bn2
    int ybf;
};
// TRALite (:122-139) on v [B][Tt][33][8]: e = mean_F v^2; y = dw conv1d k=3 causal; g = sigmoid(1x1(y))
int tra_fwd(const float* v, int B, int Tt, const float* dw_w, const float* dw_b, const float* pw_w,
            const float* pw_b, float* e, float* y, float* g, hipStream_t s, int bf = 0, const TraBn* tb = nullptr);
// out[b,t,f,2c] = v[b,t,f,c] * g[b,t,c], out[b,t,f,2c+1] = x[b,t,f,8+c]   (shuffle, :222-227), t < T
// skip (format bf; not with out2): out = the block output + skip, see bn_act's post
int gate_shuffle_fwd(const float* v, const float* g, const float* x, int B, int T, int Tt, float* out, hipStream_t s,
                     int bf = 0, float* out2 = nullptr, int out2_bf = 0, const float* skip = nullptr,
                     const TraBn* tb = nullptr);
// backward of tra + gate + shuffle: dout [B][T][33][16] -> dv [B][Tt][33][8] (written), dx channels 8..15 (written; dx_acc:
// added to what dx holds), parameter gradients written.  tmp: 3 * B*Tt*8 floats; scratch: MAX_PARTIALS * 80 floats.
int tra_gate_shuffle_bwd(const float* dout, const float* v, const float* g, const float* e, const float* y, int B,
                         int T, int Tt, const float* dw_w, const float* pw_w, float* dv, float* dx, float* d_dw_w,
                         float* d_dw_b, float* d_pw_w, float* d_pw_b, float* tmp, float* scratch, hipStream_t s,
                         int bf = 0, int dx_acc = 0, const TraBn* tb = nullptr, int gbf = 0);

// HybridLoss (loss.py:30-71).  hybrid_loss_spec: the three spectral terms -- per-workgroup sums (sum of squared
// compressed real+imag differences, sum of squared compressed-magnitude differences) into `partial`, and their
// gradient w.r.t. pred into grad (contiguous (B,257,T,2); nullptr: value only).  sisnr_terms: from the two
// waveforms, the SI-SNR term, the closed loss value (loss[0]) and, if want_grad, yp := d loss / d yp already
// divided by the iSTFT envelope (input of the iSTFT adjoint).  dwork: B * 25 doubles, coef: 2 * B floats.
int hybrid_loss_spec(const float* pred, long pb, long pf, long pt, const float* tru, long tb, long tf, long tt, int B,
                     int T, float* grad, long gb, long gf, long gt, double* partial, int* parts, hipStream_t s);
int sisnr_terms(float* yp, const float* yt, int B, long Lw, const double* spec_partial, int spec_parts, long N,
                const float* win, double* dwork, float* coef, float* loss, int want_grad, hipStream_t s);

int add(const float* a, const float* b, float* out, long n, hipStream_t s);
// out = a + b on saved tensors (n % 4 == 0); saved tensor -> fp32 copy (test taps)
int add_saved(const float* a, const float* b, float* out, long n, hipStream_t s, int bf);
int saved_to_f32(const float* src, float* dst, long n, hipStream_t s, int bf, const float* minus = nullptr);   // dst = src [- minus]

// clip_grad_norm_(max_norm) + torch.optim.Adam.step() over flat blobs of n floats in two launches (train.py:282-285):
// p / g / m / v = parameters, gradients, exp_avg, exp_avg_sq; mask[i] != 0 marks the trainable elements (others are
// left alone and do not count in the norm); max_norm <= 0: no clipping; omb1 = 1 - beta1, omb2 = 1 - beta2, step_size =
// lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) as the host computes them in double (what torch.optim.Adam does);
// out_norm (2 floats): total norm, clip coefficient.  partial: (n + 255) / 256 doubles of scratch; counter: one
// unsigned that is 0 on entry and 0 again on completion (the last-workgroup ticket of the norm).
int clip_adam(float* p, float* g, float* m, float* v, const float* mask, int n, float max_norm, float beta1, float beta2,
              float omb1, float omb2, float step_size, float bc2_sqrt, float eps, float weight_decay, float* out_norm,
              double* partial, unsigned* counter, hipStream_t s);

}  // namespace gtt
