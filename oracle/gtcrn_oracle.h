/*
 * gtcrn_oracle.h -- CPU restatement of the GTCRN-Micro hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gtcrn_micro_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / reported
 * baseline -- never as the thing measured or shipped.
 *
 * Parity status: PINNED.  Checked in tests/test_oracle_golden.py against
 * vectors produced by importing the reference itself (tests/golden/
 * make_golden.py): every stage boundary, streaming caches, the reference's
 * own causality test, its conv-wrapper test and the shipped
 * examples/noisy1.wav -> enh1.wav pair.
 *
 * All tensors fp32, row-major.  Parameter blob = the reference state_dict in
 * its own order minus the int64 num_batches_tracked entries (44 938 floats;
 * tests/golden/params_manifest.json).
 */
#ifndef GTCRN_ORACLE_H
#define GTCRN_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define GTCRN_ORACLE_NPARAMS 44938
#define GTCRN_NFFT 512
#define GTCRN_HOP 256
#define GTCRN_NBINS 257

typedef struct gtcrn_oracle gtcrn_oracle;

/* The reference's per-stream caches (gtcrn_micro_stream.py:618-623), one
 * stream.  conv[e][c][6][33] (e=0 encoder, 1 decoder), tra[e][k][8][2],
 * tcn[g][k] -> (16, 2d, 33) with d = 1,2,4,8 stored back to back (30 rows). */
typedef struct {
    float conv[2][16][6][33];
    float tra[2][3][8][2];
    float tcn[2][16 * 30 * 33]; /* block k at offset 16*33*2*(2^k-1), layout (16, 2d, 33) */
} gtcrn_oracle_state;

gtcrn_oracle *gtcrn_oracle_create(const float *params, long n_floats);
void gtcrn_oracle_destroy(gtcrn_oracle *h);

/* kind 0: sqrt-Hann periodic (infer.py:65, loss.py:50, tests); 1: Hann periodic (train.py:252). */
void gtcrn_oracle_window(int kind, float *w512);

/* Number of STFT frames for L samples, center=True: 1 + L/256 (infer.py:60-67). */
long gtcrn_oracle_num_frames(long L);

/* torch.stft(x,512,256,512,win,center=True,reflect,onesided,return_complex=False)
 * wave (B,L) -> spec (B,257,T,2). */
int gtcrn_oracle_stft(const float *wave, int B, long L, const float *win, float *spec);
/* windowed frames only (bit-exact framing/indexing check): (B,T,512) */
int gtcrn_oracle_frames(const float *wave, int B, long L, const float *win, float *frames);

/* torch.istft(view_as_complex(spec),512,256,512,win): spec (B,257,T,2) -> wave (B,256*(T-1)). */
int gtcrn_oracle_istft(const float *spec, int B, int T, const float *win, float *wave);

/* GTCRNMicro.forward, eval mode (models/gtcrn_micro.py:506-532): (B,257,T,2)->(B,257,T,2).
 * state: NULL = offline (zero history); else B states, read as the history in
 * front of frame 0 and overwritten with the history after frame T-1, i.e. with
 * T == 1 this is exactly StreamGTCRNMicro.forward (gtcrn_micro_stream.py:541-574). */
int gtcrn_oracle_forward(gtcrn_oracle *h, const float *spec, int B, int T, float *out,
                         gtcrn_oracle_state *state);

/* Stage outputs of batch item 0 of the most recent forward call.  Names follow
 * tests/golden/make_golden.py: feat erb_bm sfe en0..en4 gtcn1_b0..gtcn2_b3
 * de0..de4 erb_bs.  Returns element count (0 if unknown); copies if dst != NULL. */
long gtcrn_oracle_tap(gtcrn_oracle *h, const char *name, float *dst);

/* wave -> wave convenience used by the CPU baseline: STFT -> forward -> iSTFT. */
int gtcrn_oracle_enhance(gtcrn_oracle *h, const float *wave, int B, long L, int window_kind,
                         float *wave_out);

/* Generic causal streaming conv pieces mirrored from
 * streaming/conversion/convolution.py (used by the wrapper tests).
 * conv2d, stride 1, zero freq padding pf, time-causal via explicit history:
 * x (Cin,T,F), hist (Cin,(kt-1)*dt,F) or NULL, w (Cout,Cin/groups,kt,kf). */
int gtcrn_oracle_conv2d_causal(const float *x, const float *hist, int Cin, int T, int F, const float *w,
                               const float *b, int Cout, int kt, int kf, int dt, int df, int pf, int groups,
                               float *y /* (Cout,T,Fout) */, int *Fout);
/* ConvTranspose2d, stride 1, weight (Cin,Cout,kt,kf), freq padding pf, causal in time (first T frames). */
int gtcrn_oracle_convT2d_causal(const float *x, const float *hist, int Cin, int T, int F, const float *w,
                                const float *b, int Cout, int kt, int kf, int dt, int df, int pf,
                                float *y, int *Fout);

#ifdef __cplusplus
}
#endif
#endif
