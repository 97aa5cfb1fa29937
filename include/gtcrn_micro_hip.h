/*
 * gtcrn_micro_hip.h -- C ABI of the MI355X (gfx950) GTCRN-Micro hot path.
 *
 * The reference (bglid/GTCRN-Micro) has no FFI layer: its seam is the Python
 * class surface.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference repo).  All pointers are plain
 * caller-owned pointers; "d_" = device memory (e.g. torch.Tensor.data_ptr()),
 * "h_" = host memory.  No torch types cross this boundary.
 *
 * Every function returns 0 on success or a negative gtcrn_status; the message
 * is available from gtcrn_last_error() (thread-local).  Launches are
 * asynchronous on the given HIP stream (a hipStream_t passed as void*; NULL =
 * the default stream).  A model handle is bound to one device and is not
 * meant for concurrent calls from several threads (the reference's contract:
 * single-threaded caller, one process per GPU -- train.py:461-471).
 */
#ifndef GTCRN_MICRO_HIP_H
#define GTCRN_MICRO_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTCRN_ABI_VERSION 1
#define GTCRN_NFFT 512
#define GTCRN_HOP 256
#define GTCRN_NBINS 257
#define GTCRN_NPARAM_FLOATS 44938 /* state_dict minus num_batches_tracked */

typedef enum {
    GTCRN_OK = 0,
    GTCRN_ERR_ARG = -1,     /* bad argument (shape, null pointer, T < 1 ...) */
    GTCRN_ERR_HIP = -2,     /* a HIP runtime call failed */
    GTCRN_ERR_DEVICE = -3,  /* no gfx950 device / wrong architecture */
    GTCRN_ERR_STATE = -4    /* handle used before/after its lifetime, workspace too small under capture */
} gtcrn_status;

typedef struct gtcrn_model gtcrn_model;

int gtcrn_abi_version(void);
const char *gtcrn_last_error(void);

/* ---- parameter blob -----------------------------------------------------
 * The flat fp32 blob is the reference state_dict in its own order without the
 * int64 num_batches_tracked entries (ckpt["model"], train.py:200-216;
 * models/gtcrn_micro.py:486-504).  These describe that order so a host can
 * pack by name. */
long gtcrn_param_tensors(void);             /* number of tensors (342) */
const char *gtcrn_param_name(long i);       /* e.g. "encoder.en_convs.0.conv.weight" */
long gtcrn_param_numel(long i);
long gtcrn_param_offset(long i);            /* float offset into the blob */

/* ---- model handle -------------------------------------------------------
 * Replaces: GTCRNMicro(...).to(device) + load_state_dict(ckpt["model"]) + .eval()
 * (infer.py:37-41).  BatchNorm is folded with its running statistics
 * (eval mode).  device = HIP device ordinal. */
int gtcrn_model_create(gtcrn_model **out, const float *h_params, long n_floats, int device);
/* Re-fold after the host changed the weights (load_state_dict on a live module). */
int gtcrn_model_set_params(gtcrn_model *m, const float *h_params, long n_floats);
void gtcrn_model_destroy(gtcrn_model *m);

/* Pre-size the library-owned workspace so that later calls with batch <= B and
 * frames <= T do no allocation (required before HIP-graph capture). */
int gtcrn_model_reserve(gtcrn_model *m, int B, int T);

/* ---- windows ------------------------------------------------------------
 * kind 0: torch.hann_window(512).pow(0.5) (infer.py:65, loss.py:50, tests);
 * kind 1: torch.hann_window(512) (train.py:252).  Host helper; callers may
 * pass any 512-tap window they computed themselves. */
int gtcrn_make_window(int kind, float *h_w512);
long gtcrn_num_frames(long L); /* 1 + L/256 (torch.stft center=True) */

/* ---- STFT / iSTFT (the callers' torch.stft / torch.istft) ---------------
 * Replaces torch.stft(x,512,256,512,win,return_complex=False) (infer.py:60-67,
 * train.py:247-263): d_wave (B,L) -> d_spec.  The spectrogram is addressed as
 *   spec[b,f,t,c] = d_spec[b*sb + f*sf + t*st + c]   (c = 0 re, 1 im)
 * reference layout (B,257,T,2): sb = 257*T*2, sf = T*2, st = 2. */
int gtcrn_stft(const float *d_wave, int B, long L, const float *d_win, float *d_spec,
               long sb, long sf, long st, void *stream);
/* Windowed frames only, (B,T,512): exposes the framing/indexing for the
 * bit-exactness check (reflect pad 256, frame t = xp[256t:256t+512]). */
int gtcrn_stft_frames(const float *d_wave, int B, long L, const float *d_win, float *d_frames, void *stream);
/* Replaces torch.istft(view_as_complex(y),512,256,512,win) (infer.py:73-76):
 * d_spec (strided as above) -> d_wave (B, 256*(T-1)). */
int gtcrn_istft(const float *d_spec, long sb, long sf, long st, int B, int T, const float *d_win,
                float *d_wave, void *stream);

/* ---- offline forward ----------------------------------------------------
 * Replaces GTCRNMicro.forward(spec) in eval mode (models/gtcrn_micro.py:506-532):
 * (B,257,T,2) -> (B,257,T,2), both addressed with the strides given. */
int gtcrn_forward_spec(gtcrn_model *m, const float *d_spec_in, long isb, long isf, long ist,
                       float *d_spec_out, long osb, long osf, long ost, int B, int T, void *stream);
/* Fused caller loop of infer.py:60-76 for B equal-length clips:
 * STFT -> forward -> iSTFT; d_wave (B,L) -> d_wave_out (B, 256*(L/256)). */
int gtcrn_forward_wave(gtcrn_model *m, const float *d_wave, float *d_wave_out, int B, long L,
                       const float *d_win, void *stream);
/* The same loop for B clips of DIFFERENT lengths in one launch sequence (infer.py:48-107 takes the clips of a
 * folder one by one, whatever their lengths): row b of d_wave (B,Lmax) holds d_lengths[b] samples
 * (device int32[B], 257 <= d_lengths[b] <= Lmax); row b of d_wave_out (B, 256*(Lmax/256)) receives its
 * 256*(d_lengths[b]/256) enhanced samples -- bit-identical to gtcrn_forward_wave on that clip alone; the rest
 * of the row is left untouched.  The caller guarantees the bounds (the lengths live in device memory). */
int gtcrn_forward_wave_var(gtcrn_model *m, const float *d_wave, float *d_wave_out, int B, long Lmax,
                           const int *d_lengths, const float *d_win, void *stream);

/* ---- int8-weight / fp16-activation variant (BASELINE configs[4]) ----------------------
 * Replaces the quantised deployment path of the reference: onnx2tf -oiqt -qt per-channel -rtpo PReLU
 * (scripts/onnx2tf.sh:50-64) run by tflite_infer.py:60-107.  The reference ships no quantised model, no
 * calibration data and no TensorFlow, so this variant has its OWN stated contract -- PARITY UNPINNED:
 *   weights      every conv / linear weight of the BatchNorm-folded graph -> symmetric int8 per output channel
 *                (scale = max|w| / 127), consumed as fp16(int8 * scale); biases and PReLU slopes stay fp32
 *   activations  rounded to fp16 (round to nearest even) where a layer produces them; products and sums in fp32
 *                (v_mfma_f32_16x16x16_f16: fp16 operands, fp32 accumulate)
 *   boundary     in_scale / out_scale > 0 add the tflite model's int8 input / output tensors:
 *                x_q = clip(round(x / (scale/255)), -128, 127), x = x_q * scale/255 (tflite_infer.py:79-92 with zero
 *                point 0; the calibration convention x/scale + 0.5 in [0,1] of utils/calibration_data.py:97-106 is
 *                this quantiser, scale = 19.944473 in streaming/tflite/calib_scale.txt).  0 = fp16 boundary.
 * Same shapes/strides as gtcrn_forward_spec / gtcrn_forward_wave (STFT and iSTFT stay fp32: they are the caller's
 * torch.stft / torch.istft around the tflite interpreter, tflite_infer.py:63-101).  Offline only. */
int gtcrn_forward_spec_quant(gtcrn_model *m, const float *d_spec_in, long isb, long isf, long ist,
                             float *d_spec_out, long osb, long osf, long ost, int B, int T, float in_scale,
                             float out_scale, void *stream);
int gtcrn_forward_wave_quant(gtcrn_model *m, const float *d_wave, float *d_wave_out, int B, long L,
                             const float *d_win, float in_scale, float out_scale, void *stream);
/* Host views for CPU tests: the packed buffers with quantised weights; float -> binary16 -> float (RNE). */
int gtcrn_pack_params_quant_host(const float *h_params, long n_floats, float *h_f, int *h_i);
float gtcrn_round_to_half(float x);

/* ---- streaming ----------------------------------------------------------
 * Replaces StreamGTCRNMicro.forward(spec, conv_cache, tra_cache, tcn_cache)
 * (gtcrn_micro_stream.py:541-574) for nstreams independent streams.  The
 * per-stream state lives in device memory in the library's ring layout;
 * import/export convert from/to the reference's three caches:
 *   conv_cache (2,N,16,6,33)  tra_cache (2,3,N,8,2)
 *   tcn_cache  2 x 4 tensors (N,16,2d,33), d = 1,2,4,8, passed as 8 pointers
 *   in the order [g0 d1, g0 d2, g0 d4, g0 d8, g1 d1, ...]. */
size_t gtcrn_stream_state_bytes(void); /* per stream */
int gtcrn_stream_reset(gtcrn_model *m, void *d_state, int nstreams, void *stream);
/* nframes >= 1 consecutive frames per call: d_spec_t / d_spec_out_t are
 * (N,257,nframes,2) addressed with the strides given.  nframes == 1 (the
 * reference's loop, :626-635) is ONE kernel launch -- encoder, both GTCN stacks
 * and decoder for four streams per workgroup, nothing handed over through HBM --
 * asynchronous on `stream`, no allocation once gtcrn_model_reserve(N, 1) was
 * called: capturable into a HIP graph.  (With gtcrn_debug_enable the step runs
 * as three launches whose hand-off tensors the stage taps read.)  The state holds
 * a 16-bit frame counter; only its value mod 16 matters, it wraps freely. */
int gtcrn_stream_step(gtcrn_model *m, void *d_state, const float *d_spec_t, long isb, long isf, long ist,
                      float *d_spec_out_t, long osb, long osf, long ost, int nstreams, int nframes,
                      void *stream);
int gtcrn_stream_import(gtcrn_model *m, void *d_state, int nstreams, const float *d_conv_cache,
                        const float *d_tra_cache, const float *const *d_tcn_cache8, void *stream);
int gtcrn_stream_export(gtcrn_model *m, const void *d_state, int nstreams, float *d_conv_cache,
                        float *d_tra_cache, float *const *d_tcn_cache8, void *stream);

/* ---- standalone streaming conv wrappers -----------------------------------
 * Replaces StreamConv2d.forward / StreamConvTranspose2d.forward
 * (streaming/conversion/convolution.py:107-119, 201-253): out = conv(cat([cache, x], time)),
 * cache_out = last (kt-1)*dt rows of the input.  x (B,Cin,T,F), cache (B,Cin,(kt-1)*dt,F),
 * y (B,Cout,T,Fout); stride 1, time padding 0 (causal), frequency padding pad_f.
 * transposed = 0: Conv2d weight (Cout, Cin/groups, kt, kf);
 * transposed = 1: ConvTranspose2d weight (Cin, Cout, kt, kf) as stored by the OFFLINE module
 * (the permute + flip of convert.py:35-48 is implied).  Returns Fout (> 0) or a negative status. */
int gtcrn_stream_conv2d(const float *d_x, const float *d_cache, const float *d_w, const float *d_bias,
                        float *d_y, float *d_cache_out, int B, int Cin, int Cout, int T, int F, int kt, int kf,
                        int dt, int df, int pad_f, int groups, int transposed, void *stream);

/* ---- host-side packer view (no device needed) ---------------------------
 * The BatchNorm-folded "slot space" buffers the kernels consume, as produced
 * from a parameter blob; lets a CPU test check the weight contract
 * (convert_to_stream's permute/flip, streaming/conversion/convert.py:35-48,
 * and the shuffle renaming) without a GPU. */
void gtcrn_pack_sizes(long *n_floats, long *n_ints);
int gtcrn_pack_params_host(const float *h_params, long n_floats, float *h_f, int *h_i);

/* ---- test hooks ---------------------------------------------------------
 * Stage boundaries of the most recent gtcrn_forward_spec call, converted to
 * the reference's (C,T,F) layout and logical channel order, for batch item b.
 * Names: en0..en4, gtcn1, gtcn2, de0..de4 (de* need gtcn_debug_enable(m,1)
 * before the forward).  Returns element count, or a negative status. */
/* on = 2 (diagnostic build): phase stamps only -- a single-frame streaming step stays the ONE-launch form (its
 * stamps land in kernel slot 0 of gtcrn_debug_stamps, one row per workgroup of four streams). */
int gtcrn_debug_enable(gtcrn_model *m, int on);
/* Variable-length batches (gtcrn_forward_wave_var) of B <= 1024 utterances that are not whole rounds of 256 run the
 * per-utterance kernels in time spans: 256-workgroup rounds share the frames that exist, so a folder of a few dozen
 * files fills the chip.  Results are bit-identical either way; on = 0 goes back to one workgroup per utterance (the A/B
 * switch of tests and measurements).  Default on. */
int gtcrn_var_spans_enable(gtcrn_model *m, int on);
/* Single-frame streaming steps (gtcrn_stream_step with nframes == 1): form 0 (default) runs the whole step as ONE
 * kernel, nothing handed over through HBM -- four streams per workgroup (k_stream_ms) while one round of workgroups
 * covers the streams, seven per workgroup (k_stream_wide: eight waves x two tiles, parameters streamed through LDS)
 * once the stream count fills the chip more than once; form 1 runs the three-launch form (encoder, both GTCN stacks,
 * decoder; hand-off tensors in HBM) that the stage taps use; forms 2 and 3 pin the one-launch step to four / seven
 * streams per workgroup whatever the count.  Bit-identical outputs and ring state in every form
 * (tests/test_gpu_stream.py); the A/B switch of the capacity measurements. */
int gtcrn_stream_form(gtcrn_model *m, int form);
/* Which one-launch form the default (form 0) runs for `nstreams` single-frame steps: the streams per workgroup, 4
 * (k_stream_ms) or 7 (k_stream_wide).  Pure host logic (no device is touched): both forms run one workgroup per CU, so a
 * step costs rounds-of-256-workgroups x the form's time per round (32.9 us against ~1.5 x that); the wide form is taken
 * once the narrow one needs more rounds than it. */
int gtcrn_stream_streams_per_workgroup(int nstreams);
long gtcrn_debug_tap(gtcrn_model *m, const char *name, int b, float *h_dst, long cap);
/* Diagnostic build only (libgtcrn_micro_hip_stamps.so, -DGT_STAMPS): per-workgroup sums of shader
 * cycles spent in each barrier-delimited phase of kernel 0 encoder, 1 gtcn1, 2 gtcn2, 3 decoder,
 * (B,16) values.  The product library returns zeros. */
long gtcrn_debug_stamps(gtcrn_model *m, int kernel, unsigned long long *h_dst, long cap);
/* Checks the MFMA f32 16x16x4 lane maps the kernels rely on (exact integer
 * data, asymmetric operands).  0 = as assumed. */
/* 16-bit PCM at the host boundary: the reference's callers read mono 16-bit WAV files with soundfile (infer.py:54: the
 * samples arrive as int16 / 32768) and write the enhanced waveform back as 16-bit PCM (infer.py:113, sf.write).  A caller
 * that hands the int16 SAMPLES over moves half the bytes across the host link, which is what bounds a served pipeline
 * (bench.py io.served_pcm16_*).  gtcrn_pcm16_to_f32: x = s / 32768 (exact); gtcrn_f32_to_pcm16: s = clip(rint(y * 32768),
 * -32768, 32767), round half to even.  Device pointers, 16-byte aligned; n samples, a multiple of 8; asynchronous on
 * `stream`.  The bulk offline driver (gtcrn_micro_amd/infer.py) uses both. */
int gtcrn_pcm16_to_f32(int device, const short *d_pcm, float *d_wave, long n, void *stream);
int gtcrn_f32_to_pcm16(int device, const float *d_wave, short *d_pcm, long n, void *stream);
int gtcrn_selftest_mfma(int device);
/* The exact three-way bf16 split the dense 3x3 runs on (kernels.hip split3 / join3 / split_mm6), on caller-chosen
 * values: h_x[n] (n a multiple of 4) -> h_planes[3][n] (hi, mid, lo as floats) and h_joined[n] (= h_x bit for bit
 * wherever all three planes are normal numbers); optionally h_A (16x32, row major) * h_B (32x16) -> h_D (16x16)
 * through the same six-product helper the kernels use, from operands split on the device.  Host pointers. */
int gtcrn_selftest_split3(int device, const float *h_x, long n, float *h_planes, float *h_joined, const float *h_A,
                          const float *h_B, float *h_D);
/* HIP-event timing of every kernel launch (events recorded on the call's stream, no
 * synchronisation inside the timed region).  gtcrn_timing_enable(m,1) clears the record;
 * gtcrn_timing_read returns, for kernel idx in [0, gtcrn_timing_kernels()) -- every timed launch records which
 * kernel it was, so offline and streaming calls keep their own rows (k_front, k_encoder_gt, k_gtcn1, k_gtcn2,
 * k_decoder, k_istft offline; k_stream_ms / k_stream_wide for single-frame streaming steps; ...) -- its name, the average device time
 * in ms over the launches recorded since and their count.  idx = -1 - k only returns the name of kernel k (m may be
 * NULL).  on = 2 + idx records events around kernel idx ONLY: every event pair costs a few microseconds of dispatch
 * gap, so the timed region of bench.py keeps just the dominant kernel's.  Used for the roofline line. */
int gtcrn_timing_enable(gtcrn_model *m, int on);
int gtcrn_timing_kernels(void);
int gtcrn_timing_read(gtcrn_model *m, int idx, char *name, int name_cap, float *ms, int *launches);

/* ---- train step (model forward/backward) ---------------------------------
 * Replaces: `enhanced = self.model(noisy_spec)` with the module in .train() mode and the model part
 * of `loss.backward()` (train.py:265, 280; models/gtcrn_micro.py:506-532).  Every nn.BatchNorm2d
 * normalises with the statistics of the batch and updates its running estimates in place
 * (momentum 0.1, unbiased variance), exactly as nn.BatchNorm2d.train() does; the transposed
 * depth convs of the decoder produce T+2 frames whose tail takes part in the statistics
 * (models/gtcrn_micro.py:238-251).  The loss, the optimiser and the scheduler stay with the caller
 * (PyTorch; train.py:267-288).
 *
 * d_params: the canonical blob (GTCRN_NPARAM_FLOATS floats, raw, NOT folded) in device memory; the
 *   forward writes the new running_mean/running_var into it.
 * d_grads:  same layout; the backward writes d loss / d parameter for the 248 trainable tensors and
 *   zeros in the slots of buffers (running statistics, ERB filterbank).
 * Spectrograms are addressed by strides like gtcrn_forward_spec.  The backward differentiates the
 * most recent forward of this trainer (same d_params, same d_spec).  One trainer per device,
 * single caller thread (train.py:461-471); no CPU fallback. */
typedef struct gtcrn_trainer gtcrn_trainer;
int gtcrn_trainer_create(gtcrn_trainer **out, int device);
void gtcrn_trainer_destroy(gtcrn_trainer *t);
long gtcrn_train_workspace_bytes(int B, int T);   /* saved activations + gradient buffers (fp32 storage) */
/* Storage of the SAVED activations (everything the backward re-reads): 0 = fp32, the reference's own precision
 * (train.py:239-288 trains in fp32); 1 = bf16 (BASELINE configs[3] asks for bf16: half the bytes of every pass of the
 * HBM-bound layer-at-a-time step; the forward then IS the bf16-activation network: its consumers read the rounded
 * values); 4 = bf16 SAVES with an exact forward chain: every forward tensor is written twice -- the fp32 value the next
 * layer reads and the bf16 copy the backward re-reads -- so the output equals mode 0's bit for bit and the gradient
 * differs from mode 0's only by the rounding of the saved tensors, at mode 1's workspace (the fp32 buffers of the chain
 * are short-lived and share the backward's scratch region); 5 = mode 1 with the gradient tensors handed from one
 * unit's backward to the next stored in bf16 as well (what bf16 autocast training keeps: the forward is mode 1's bit for
 * bit; a unit rounds the gradient it hands on where it stores it).  Arithmetic, BatchNorm statistics and reductions,
 * every PARAMETER gradient, the gradient all-reduce, Adam and the master weights stay fp32 in all modes.  Takes effect
 * at the next forward. */
int gtcrn_trainer_set_storage(gtcrn_trainer *t, int storage);
/* Workspace of a (B, T) problem in `storage` with the DEFAULT fusion mask (every pass fusion on).  A trainer whose mask
 * was changed with gtcrn_trainer_set_fusions stores more tensors (about 6 GiB more at B = 512 with mask 7):
 * gtcrn_trainer_workspace_bytes plans with the trainer's own storage mode and mask. */
long gtcrn_train_workspace_bytes2(int B, int T, int storage);
long gtcrn_trainer_workspace_bytes(gtcrn_trainer *t, int B, int T);
/* Diagnostic: which pass fusions of the train step are active (default: all).  bit 0: BatchNorm + PReLU of a unit
 * applied by the conv that consumes it (normalise-on-load; the forward is bit-identical with and without), bit 1: the
 * depthwise unit's backward in one pass, bit 2: BatchNorm reductions accumulated by the kernel that produces their
 * gradient input, bit 3 (needs bits 0 and 2): an activation whose only readers are a normalise-on-load conv and that
 * conv's fused backward is not stored -- the backward recomputes it from the conv output it reads anyway (22 of the 46
 * units: one tensor write less in the forward, one read less in the backward, 6 GiB less workspace at B = 512).
 * bit 4: the two gradients every encoder output receives (decoder skip + main path) are summed by accumulating stores
 * of the kernels that produce the second one, not by five add passes.  bit 5 (needs bit 2): the reductions of
 * point_conv1 (six blocks) and en_convs.0 ride in the adjoint conv that produces their gradient input.  bit 6: the
 * backward of the encoder's depthwise 3x3 unit (dy, weight gradient, data gradient) in one LDS-tiled pass.  bit 7: the
 * same for the decoder's dense transposed 3x3 unit (both matrix products from LDS images of dy and x).  bit 8 (needs bit
 * 0): point_conv1's BatchNorm + PReLU applied by LDS-tiled depth convs while they stage their input tile.  bit 9: the
 * backward of the two 16 -> 16 (1,5) stride-2 units (en_convs.1, de_convs.3) from LDS tiles.  bit 10: the second
 * stage of every BatchNorm reduction (forward statistics + running estimates; backward means, dgamma, dbeta, dslope) runs
 * in the LAST workgroup of the kernel that produces the per-workgroup sums (two-level last-arriver reduction, agent-scope
 * write-through hand-off, fixed summation order) instead of 92 one-workgroup finish launches per step.  bit 11 (not
 * in storage mode 4): each of the decoder's five sums x + en_outs[..] (models/gtcrn_micro.py:463-469) is written by the layer
 * that produces x (the last TCN block's normalise pass, the decoder blocks' gate/shuffle, de_convs.3's normalise pass)
 * instead of an add pass; x itself is not stored (its test tap is sum - skip); in the 16-bit modes x is rounded to
 * the storage format before the add, as the stored x was, and the backward no longer recomputes the sums.  bit 12 (needs bit 0; not in storage
 * mode 4): point_bn2 -- the one BatchNorm with no activation behind it -- is applied on load by its four readers
 * (TRALite's energy, the gate/shuffle, their two backward passes): six normalise passes per step and the tensor they
 * wrote are gone, the values are the same bit for bit.  bit 13: the 28 pointwise forward convs of a step run in a
 * dedicated kernel (flat positions, compile-time formats, two tiles per iteration with the next two requested) instead of
 * the general strided / padded conv kernel (and the two 16 -> 16 (1,5) stride-2 layers in one of their own, k_c15_fwd).  bit 14: the TCN's dilated depthwise (3,1) forward in a column form -- a
 * thread walks one residue class of frames modulo the dilation, so every input is normalised once instead of three
 * times and a chunk's loads are issued together.  Both are bit-identical to the kernels they replace (conv outputs;
 * the BatchNorm statistics to the float); bit 14 also runs that unit's fused backward in the column form (dy and the
 * recomputed activation once per element; sums in another order).  bit 15: the weight-gradient finishes of a backward
 * pass (44 small launches) are recorded and run as two batched launches at its end, partial sums in a pool (same sums,
 * same order: bit-identical gradients).  Default 65535.
 * 0 runs the layer-at-a-time passes (tests/test_gpu_train.py compares them).  Takes effect at the next forward; not
 * part of the reference's interface. */
int gtcrn_trainer_set_fusions(gtcrn_trainer *t, int mask);
int gtcrn_train_forward(gtcrn_trainer *t, float *d_params, const float *d_spec, long sb, long sf, long st,
                        float *d_out, long ob, long of, long ot, int B, int T, void *stream);
int gtcrn_train_backward(gtcrn_trainer *t, const float *d_params, const float *d_spec, long sb, long sf,
                         long st, const float *d_grad_out, long gb, long gf, long gt, float *d_grads,
                         void *stream);
/* HybridLoss (loss.py:30-71): 30 * (MSE of the 0.3-compressed real and imaginary parts) + 70 * MSE of the
 * compressed magnitudes + SI-SNR of the sqrt-Hann iSTFTs, batch mean.  d_loss receives one float; d_grad
 * (optional) the gradient w.r.t. d_pred as a contiguous (B,257,T,2) tensor.  Replaces
 * `loss = self.loss_func(enhanced, clean_spec)` and the loss part of `loss.backward()` (train.py:267, 280). */
int gtcrn_train_loss(gtcrn_trainer *t, const float *d_pred, long pb, long pf, long pt, const float *d_true, long tb,
                     long tf, long tt, int B, int T, float *d_loss, float *d_grad, void *stream);
/* The same with the gradient addressed by strides (gb, gf, gt) like the spectrograms: a caller that keeps its
 * spectrograms frame-major -- (B,257,T,2)-shaped views of (B,T,257,2) memory, which every kernel of the step then reads
 * and writes in 2 KB rows -- gets the gradient in that layout too. */
int gtcrn_train_loss_strided(gtcrn_trainer *t, const float *d_pred, long pb, long pf, long pt, const float *d_true,
                             long tb, long tf, long tt, int B, int T, float *d_loss, float *d_grad, long gb, long gf,
                             long gt, void *stream);
/* clip_grad_norm_ + Adam over flat blobs in two launches.  Replaces `torch.nn.utils.clip_grad_norm_(self.model.parameters(),
 * clip_grad_norm_value)` + `self.optimizer.step()` (train.py:282-285; Adam(lr) from train.py:90, conf/cfg_train_DNS3.yaml
 * clip 3.0) for a model whose parameters, gradients and Adam moments are views of four blobs in the canonical layout
 * (PyTorch walks the 248 views: ~300 launches per step): total 2-norm of the masked gradient (double accumulation,
 * fixed order), gradients scaled in place by min(max_norm / (norm + 1e-6), 1) (max_norm <= 0: no clipping), then
 * torch.optim.Adam's update (no amsgrad; L2 weight_decay added to the gradient; bias corrections from `step` >= 1 in
 * double) of every element with d_mask != 0.  d_norm_out (optional, 2 floats): the total norm clip_grad_norm_ returns,
 * and the clip coefficient.  d_workspace: gtcrn_clip_adam_workspace_bytes(n) bytes of device memory, zeroed ONCE by the
 * caller (it holds the norm's last-workgroup ticket, which every call leaves at zero).  The learning-rate schedule stays
 * with the caller (utils/scheduler.py is host scalar arithmetic).  Asynchronous on `stream`. */
long gtcrn_clip_adam_workspace_bytes(long n);
int gtcrn_clip_adam_step(int device, float *d_params, float *d_grads, float *d_exp_avg, float *d_exp_avg_sq,
                         const float *d_mask, long n, float max_norm, double lr, double beta1, double beta2, double eps,
                         double weight_decay, long step, float *d_norm_out, void *d_workspace, void *stream);
/* Test hook: train-mode activation of the most recent forward at a stage boundary (en0..en4, gtcn1,
 * gtcn2, de0..de4), channels-last (B, T, F, C) in the reference's channel order; shape4 receives
 * the four extents; d_out may be NULL to query the shape. */
int gtcrn_train_tap(gtcrn_trainer *t, const char *name, float *d_out, long *shape4, void *stream);

#ifdef __cplusplus
}
#endif
#endif
