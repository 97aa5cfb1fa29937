// train.cpp -- the train-mode forward/backward of GTCRN-Micro behind the C ABI (gtcrn_train_* in
// include/gtcrn_micro_hip.h).  Host sequencing only; the arithmetic is in train_kernels.hip.
//
// Reference: GTCRNMicro.forward with every nn.BatchNorm2d in .train() mode (models/gtcrn_micro.py:
// 506-532) and autograd's backward of it, as driven by train.py:239-288.  The model is a chain of
// "units" conv -> BatchNorm(batch statistics) -> activation; the forward keeps, per unit, the conv
// output y and the activation a (the next unit's input), which is all the backward needs
// (z and xhat are recomputed from y and the saved statistics).
#include "../../include/gtcrn_micro_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"
#include "pack.h"
#include "train_kernels.h"

extern "C" void gtcrn_set_error_(const char* msg);   // api.cpp: thread-local last error
extern "C" int gtcrn_device_twiddles_(float** out);  // api.cpp: per-device FFT twiddles

namespace {

using gtt::ConvGeom;
using gtt::DwGeom;

int tfail(int code, const std::string& m) {
    gtcrn_set_error_(m.c_str());
    return code;
}
#define T_HIP(expr)                                                                                  \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return tfail(GTCRN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
// pool of the deferred weight-gradient partials (fusion bit 15): 28 pointwise units x 1024 x 272 floats + the 3x3 / 1x5 /
// depthwise units = 14.3 M floats at B = 512; a unit that does not fit finishes right away as before
constexpr size_t WFIN_POOL_FLOATS = (size_t)16 << 20;
#define T_RUN(expr)                                                                                  \
    do {                                                                                             \
        int e_ = (expr);                                                                             \
        if (e_ != 0) return tfail(GTCRN_ERR_HIP, std::string(#expr) + ": " +                         \
                                  hipGetErrorString(static_cast<hipError_t>(e_)));                   \
    } while (0)

struct Unit {                 // conv (dense or depthwise) + BatchNorm + activation
    bool dw = false;
    ConvGeom cg{};
    DwGeom dg{};
    long o_w = -1, o_b = -1, o_bn = -1, o_slope = -1;   // float offsets into the canonical blob
    int act = gtt::ACT_NONE;
    int C = 16;               // channels of y
    long n = 0;               // positions of y
    const float* x = nullptr; // input tensor
    float* y = nullptr;       // conv output (pre-BN)
    float* a = nullptr;       // activation output
    const float* res = nullptr;
    float* stats = nullptr;   // mean[C], invstd[C]
    // "bf16 saves, exact chain" storage (gtcrn_trainer_set_storage 4): x, y, a, res above are the 16-bit copies the
    // BACKWARD reads; the forward runs on the fp32 twins below (short-lived buffers of the chain pool, see plan()) and
    // writes both.  ac == nullptr: nobody but a fused consumer reads this activation in fp32.  bstats: what the
    // backward takes as the unit's statistics (== stats except in that mode: the mean of the centred 16-bit copy)
    const float* xc = nullptr;
    float* yc = nullptr;
    float* ac = nullptr;
    const float* resc = nullptr;
    const float* bstats = nullptr;
    // normalise-on-load (gtt::BnPre): `pre` = the unit in front whose BatchNorm + PReLU this unit's conv applies while
    // loading that unit's y (and whose activation it writes); `deferred` = this unit's own bn_act pass is left to its
    // consumer.  Set for conv1 -> conv2 -> conv3 of the TCN blocks and depth_conv -> point_conv2 of the GTConv blocks.
    Unit* pre = nullptr;
    bool deferred = false;
    // backward: `front` = the unit whose gradient input is this unit's dx (TCN conv2 -> conv1): the fused depthwise
    // backward accumulates that unit's BatchNorm reduction too (gtt::dwunit_bwd)
    Unit* front = nullptr;
    // `lean`: this unit's activation has ONE forward reader (a normalise-on-load conv, which forms it from y) and one
    // backward reader (that conv's fused backward, which recomputes it from y too: gtt::DwUnitNext::recompute_x), so it
    // is never stored -- a == nullptr.  Set by plan() for conv1 / conv2 of the TCN blocks and depth_conv of the GTConv
    // blocks when fusion bits 1, 4 (and 2 for conv1) and 8 are on.
    bool lean = false;
    // fusion bit 11: `a` holds activation + post (the next decoder layer's input x + skip, written by this unit's bn_act);
    // the plain activation is not stored
    const float* post = nullptr;
};

struct GtBlock {              // GTConvBlock (models/gtcrn_micro.py:167-253)
    Unit pc1, depth, pc2;
    long o_tra = -1;          // tra.depth_conv.weight; then .bias, point_conv.weight, .bias
    int Tt = 0;               // frames of the depth/pc2/TRA tensors (T, or T+2 for the transposed conv)
    const float* xin = nullptr;  // block input (after the skip add in the decoder)
    float* s = nullptr;       // decoder: x + skip
    float *e = nullptr, *yt = nullptr, *g = nullptr, *out = nullptr;
    const float* xinc = nullptr; // exact chain: fp32 twins of xin / s / out
    float *sc = nullptr, *outc = nullptr;
    const float* post = nullptr; // fusion bit 11: `out` holds block output + post (see Unit::post)
    bool bn2_load = false;       // fusion bit 12: TRALite and the gate/shuffle apply point_bn2 while loading point_conv2's
                                 // output; pc2.a is not stored and its normalise pass does not run
};

struct TcnBlock {             // TCN (models/gtcrn_micro.py:256-310)
    Unit c1, c2, c3;
};

}  // namespace

struct gtcrn_trainer {
    int device = 0;
    int bf = 0;                   // format of the saved activations / block outputs: 0 fp32, 1 bf16
    int ybf = 0;                  // format of the saved conv outputs in front of a BatchNorm: 0 fp32, 1 bf16, 2 fp16
    int exact = 0;                // 1: bf16 SAVES only -- the forward chain itself runs in fp32 (storage code 4)
    int gbf = 0;                  // 1: the gradient tensors handed between units are bf16 as well (storage code 5, with bf = ybf = 1)
    float *ebc = nullptr, *f0c = nullptr, *s3c = nullptr, *s4c = nullptr;   // fp32 twins of eb, f0, s3, s4 (exact chain)
    const float *dec_in[3] = {nullptr, nullptr, nullptr};                   // ... of the decoder blocks' first addends
    int B = 0, T = 0;
    bool planned = false, have_fwd = false;
    bool shift_ready = false;     // bf16 storage: the per-unit centring shifts hold a previous step's batch means
    float* arena = nullptr;
    size_t arena_floats = 0;
    float* fscratch = nullptr;    // wgrad / TRA partial sums
    float* wfin_pool = nullptr;   // fusion bit 15: every unit's weight-gradient partials until the batched finish (WFIN_POOL_FLOATS)
    double* dscratch = nullptr;   // BatchNorm partial sums
    double* fin_gpart = nullptr;  // in-launch finish of the BatchNorm reductions (fusion bit 10): group sums and
    unsigned* fin_ctr = nullptr;  // arrival counters (zero between launches), see train_kernels.h
    int fusions = 65535;                // gtcrn_trainer_set_fusions: 1 normalise-on-load, 2 depthwise backward, 4 riding reductions,
                                      // 8 single-reader activations recomputed in the backward instead of stored,
                                      // 16 skip gradients accumulated in place (no add passes in the backward),
                                      // 32 reductions riding in the adjoint convs of the 3x3 units and of en_convs.1,
                                      // 64 the depthwise 3x3 unit's backward in one LDS-tiled pass, 128 the dense 3x3 unit's,
                                      // 256 point_conv1's BatchNorm + PReLU applied by the LDS-tiled depth convs while staging,
                                      // 512 the backward of en_convs.1 / de_convs.3 from LDS tiles,
                                      // 1024 the second stage of every BatchNorm reduction in the last workgroup of the
                                      // kernel that produces its partial sums (no finish launches),
                                      // 2048 (not the exact chain) the decoder's sums x + skip written by the layer that produces x,
                                      // 4096 point_bn2 applied on load by TRALite / gate-shuffle (forward and backward),
                                      // 8192 the pointwise forward convs in their dedicated kernel (k_pw_fwd),
                                      // 16384 the TCN's dilated depthwise unit (forward and backward) in the column form,
                                      // 32768 the weight-gradient finishes of a backward pass recorded and run as a batch
    const void* red_unit = nullptr;   // backward: the unit whose BatchNorm reduction already sits in dscratch ...
    int red_parts = 0;                // ... as this many per-workgroup partial sums (see unit_bwd)
    std::map<std::string, long> off;   // parameter name -> blob offset
    // plan
    float *eb = nullptr, *f0 = nullptr;
    Unit en0, en1, de3, de4;
    GtBlock enc[3], dec[3];
    TcnBlock tcn[8];
    float *s3 = nullptr, *s4 = nullptr, *s_tmp = nullptr;
    bool share_sums = false;
    // backward buffers
    float *dm = nullptr, *gs0 = nullptr, *gs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float *q1 = nullptr, *q2 = nullptr, *dy = nullptr, *dv = nullptr, *dhd = nullptr, *dh = nullptr, *tmp_tra = nullptr;
    float *d65 = nullptr, *df0 = nullptr;
    std::map<std::string, std::pair<const float*, std::vector<int>>> taps;  // name -> (ptr, {T', F, C})
    std::map<std::string, const float*> tap_minus;   // fusion bit 11: the tap is ptr - this (the stored tensor is a sum)
    bool fuse_sums = false;
    // HybridLoss workspace: the two iSTFT waveforms, the sqrt-Hann window, SI-SNR coefficients
    float* loss_ws = nullptr;
    size_t loss_ws_floats = 0;
    float* d_win = nullptr;
};

namespace {

struct Bump {
    size_t used = 0;              // in floats
    float* base = nullptr;
    int bf = 0, ybf = 0, exact = 0;
    float* take(size_t n) {
        n = (n + 63) & ~size_t(63);
        float* p = base ? base + used : nullptr;
        used += n;
        return p;
    }
    // a SAVED tensor of n elements (fp32 or bf16 per the trainer's storage); handled through a float* either way
    int gbf = 0;
    float* take_saved(size_t n) { return take(bf ? (n + 1) / 2 : n); }
    size_t gsz(size_t n) const { return gbf ? (n + 1) / 2 : n; }       // floats of an inter-unit gradient tensor of n elements
    float* take_grad(size_t n) { return take(gsz(n)); }
    float* take_saved_y(size_t n) { return take(ybf ? (n + 1) / 2 : n); }
};

ConvGeom conv_geom(int B, int T, int Tout, int Fin, int Fout, int CinT, int cin_off, int Cin, int CoutT, int Cout,
                   int nkt, int nkf, int t0, int t1, int t2, int f_mode, int sf, int pf, int w_co, int w_ci, int w_kt,
                   int w_kf) {
    ConvGeom g{};
    g.B = B; g.Tin = T; g.Tout = Tout; g.Fin = Fin; g.Fout = Fout;
    g.CinT = CinT; g.cin_off = cin_off; g.Cin = Cin; g.CoutT = CoutT; g.cout_off = 0; g.Cout = Cout;
    g.nkt = nkt; g.nkf = nkf; g.t_off[0] = t0; g.t_off[1] = t1; g.t_off[2] = t2;
    g.f_mode = f_mode; g.sf = sf; g.pf = pf; g.w_co = w_co; g.w_ci = w_ci; g.w_kt = w_kt; g.w_kf = w_kf;
    g.accumulate = 0;
    g.in_bf = g.out_bf = 0;
    return g;
}
ConvGeom adjoint(const ConvGeom& g, int accumulate) {
    ConvGeom a = g;
    a.Tin = g.Tout; a.Tout = g.Tin; a.Fin = g.Fout; a.Fout = g.Fin;
    a.CinT = g.CoutT; a.cin_off = g.cout_off; a.Cin = g.Cout;
    a.CoutT = g.CinT; a.cout_off = g.cin_off; a.Cout = g.Cin;
    for (int k = 0; k < 3; ++k) a.t_off[k] = -g.t_off[k];
    a.f_mode = 1 - g.f_mode;
    a.w_co = g.w_ci; a.w_ci = g.w_co;
    a.accumulate = accumulate;
    a.in_bf = a.out_bf = 0;       // gradients are fp32
    return a;
}
DwGeom adjoint(const DwGeom& g, int accumulate) {
    DwGeom a = g;
    a.Tin = g.Tout; a.Tout = g.Tin;
    for (int k = 0; k < 3; ++k) { a.t_off[k] = -g.t_off[k]; a.f_off[k] = -g.f_off[k]; }
    a.accumulate = accumulate;
    a.in_bf = a.out_bf = 0;
    return a;
}

long P(gtcrn_trainer* t, const std::string& name) {
    auto it = t->off.find(name);
    return it == t->off.end() ? -1 : it->second;
}

void unit_params(gtcrn_trainer* t, Unit& u, const std::string& conv, const std::string& bn, const std::string& act) {
    u.o_w = P(t, conv + ".weight");
    u.o_b = P(t, conv + ".bias");
    u.o_bn = P(t, bn + ".weight");
    u.o_slope = act.empty() ? -1 : P(t, act + ".weight");
}

void alloc_unit(Bump& b, Unit& u, long n, int C, bool lean = false) {
    u.n = n; u.C = C;
    u.lean = lean;
    u.y = b.take_saved_y((size_t)n * C);
    u.a = lean ? nullptr : b.take_saved((size_t)n * C);
    u.stats = b.take(128);              // mean[C], invstd[C]; +32: the centring shift of the stored y (bf16 storage);
                                        // +64: mean - shift, invstd for the backward (exact chain, see bn_stats)
    u.bstats = b.exact ? u.stats + 64 : u.stats;
    u.cg.in_bf = u.dg.in_bf = b.bf;     // forward geometry: saved activation in -> saved conv output
    u.cg.out_bf = u.dg.out_bf = b.ybf;
}

// The fp32 buffers of the exact chain live only from their producer to their last forward reader, so they come from a
// small pool with explicit lifetimes (first fit, coalescing free list), laid over the region of the backward's scratch
// tensors: nothing of the chain is needed once the forward is done, nothing of the backward's before it starts.
struct ChainPool {
    std::vector<std::pair<size_t, size_t>> fr;     // free blocks (offset, size), sorted by offset
    size_t hi = 0;
    static size_t rnd(size_t n) { return (n + 63) & ~size_t(63); }
    size_t get(size_t n) {
        n = rnd(n);
        for (size_t i = 0; i < fr.size(); ++i)
            if (fr[i].second >= n) {
                const size_t o = fr[i].first;
                if (fr[i].second == n) fr.erase(fr.begin() + i);
                else { fr[i].first += n; fr[i].second -= n; }
                return o;
            }
        if (!fr.empty() && fr.back().first + fr.back().second == hi) {      // grow the last free block
            const size_t o = fr.back().first;
            hi = o + n;
            fr.pop_back();
            return o;
        }
        const size_t o = hi;
        hi += n;
        return o;
    }
    void put(size_t o, size_t n) {
        n = rnd(n);
        size_t i = 0;
        while (i < fr.size() && fr[i].first < o) ++i;
        fr.insert(fr.begin() + i, {o, n});
        if (i + 1 < fr.size() && fr[i].first + fr[i].second == fr[i + 1].first) { fr[i].second += fr[i + 1].second; fr.erase(fr.begin() + i + 1); }
        if (i > 0 && fr[i - 1].first + fr[i - 1].second == fr[i].first) { fr[i - 1].second += fr[i].second; fr.erase(fr.begin() + i); }
    }
};

// lays out every tensor of one (B, T) problem in the arena; base == nullptr: size query
size_t plan(gtcrn_trainer* t, int B, int T, float* base) {
    Bump b;
    b.base = base;
    b.bf = t->bf;
    b.ybf = t->ybf;
    b.exact = t->exact;
    b.gbf = t->gbf;
    const long n129 = (long)B * T * 129, n65 = (long)B * T * 65, n33 = (long)B * T * 33;
    const int T2 = T + 2;
    const long n33x = (long)B * T2 * 33;
    const bool fuse = t->ybf <= 1 && (t->fusions & 1);      // (not for the fp16 diagnostic storage)
    // activations whose only readers are a normalise-on-load conv and that conv's fused backward are not stored (the
    // conditions under which unit_bwd takes those fused forms with a riding reduction)
    const bool lean = fuse && t->bf == t->ybf && (t->fusions & 4) && (t->fusions & 8);
    const bool lean_c1 = lean && (t->fusions & 2);
    // fusion bit 11 (16-bit storage: the sums then get buffers of their own instead of ONE shared buffer that the
    // backward refills by recomputing each sum -- the same memory, the block outputs' buffers go): a sum is written by the layer that
    // PRODUCES x -- the last TCN block's bn_act, the decoder blocks' gate/shuffle, de_convs.3's bn_act -- instead of a pass
    // that reads x and the skip and writes the sum; x itself (an output nobody else reads) is not stored
    t->fuse_sums = (t->fusions & 2048) && !t->exact;
    t->taps.clear();
    t->tap_minus.clear();
    {   // (the fp32 twins of the exact chain are assigned at the end; a unit that has none must not keep an old one)
        Unit* all[4 + 6 * 3 + 8 * 3] = {&t->en0, &t->en1, &t->de3, &t->de4};
        int n = 4;
        for (int k = 0; k < 3; ++k)
            for (GtBlock* g : {&t->enc[k], &t->dec[k]}) { all[n++] = &g->pc1; all[n++] = &g->depth; all[n++] = &g->pc2; }
        for (int i = 0; i < 8; ++i) { all[n++] = &t->tcn[i].c1; all[n++] = &t->tcn[i].c2; all[n++] = &t->tcn[i].c3; }
        for (int i = 0; i < n; ++i) { all[i]->xc = nullptr; all[i]->yc = nullptr; all[i]->ac = nullptr; all[i]->resc = nullptr; }
    }
    t->eb = b.take_saved(n129 * 3);
    t->f0 = b.take_saved(n129 * 3);
    // encoder.en_convs.0/1: ConvBlock (models/gtcrn_micro.py:344-364)
    t->en0.cg = conv_geom(B, T, T, 129, 65, 3, 0, 3, 16, 16, 1, 5, 0, 0, 0, 0, 2, 2, 15, 5, 5, 1);
    unit_params(t, t->en0, "encoder.en_convs.0.conv", "encoder.en_convs.0.bn", "encoder.en_convs.0.act");
    t->en0.act = gtt::ACT_PRELU; t->en0.x = t->f0;
    alloc_unit(b, t->en0, n65, 16);
    t->en1.cg = conv_geom(B, T, T, 65, 33, 16, 0, 16, 16, 16, 1, 5, 0, 0, 0, 0, 2, 2, 80, 5, 5, 1);
    unit_params(t, t->en1, "encoder.en_convs.1.conv", "encoder.en_convs.1.bn", "encoder.en_convs.1.act");
    t->en1.act = gtt::ACT_PRELU; t->en1.x = t->en0.a;
    alloc_unit(b, t->en1, n33, 16);
    // en_convs.1's adjoint produces en_convs.0's gradient input -- the TOTAL one only when it accumulates on top of the
    // skip's part (fusion bit 4)
    t->en1.front = (t->fusions & 16) ? &t->en0 : nullptr;
    t->taps["en0"] = {t->en0.a, {T, 65, 16}};
    t->taps["en1"] = {t->en1.a, {T, 33, 16}};
    const float* X = t->en1.a;
    const float* skips[5] = {t->en0.a, t->en1.a, nullptr, nullptr, nullptr};
    auto gt_block = [&](GtBlock& k, const std::string& p, bool deconv, const float* xin) {
        const int Tt = deconv ? T2 : T;
        const long nt = deconv ? n33x : n33;
        k.Tt = Tt; k.xin = xin;
        // point_conv1: Conv2d(8,16,1) weight [16][8] / ConvTranspose2d(8,16,1) weight [8][16]
        k.pc1.cg = conv_geom(B, T, T, 33, 33, 16, 0, 8, 16, 16, 1, 1, 0, 0, 0, 0, 1, 0, deconv ? 1 : 8, deconv ? 16 : 1, 1, 1);
        unit_params(t, k.pc1, p + ".point_conv1", p + ".point_bn1", p + ".point_act");
        k.pc1.act = gtt::ACT_PRELU; k.pc1.x = xin;
        // point_conv1's activation: read once by the LDS-tiled depth conv (from y, fusion bit 8) and once by that conv's
        // fused backward, which stages it from y as well (bits 6 / 7): never stored then
        const bool lean_pc1 = lean && (t->fusions & 256) && (t->fusions & (deconv ? 128 : 64));
        alloc_unit(b, k.pc1, n33, 16, lean_pc1);
        if (deconv) {
            // depth_conv: ConvTranspose2d(16,16,(3,3),padding=(0,1)), weight [in][out][kt][kf]; T+2 output frames
            k.depth.cg = conv_geom(B, T, T2, 33, 33, 16, 0, 16, 16, 16, 3, 3, 0, -1, -2, 1, 1, 1, 9, 144, 3, 1);
        } else {
            // depth_conv: Conv2d(16,16,(3,3),padding=(0,1),groups=16) on the input padded by 2 frames in front
            k.depth.dw = true;
            DwGeom d{};
            d.B = B; d.Tin = T; d.Tout = T; d.F = 33; d.C = 16; d.nkt = 3; d.nkf = 3;
            d.t_off[0] = -2; d.t_off[1] = -1; d.t_off[2] = 0; d.f_off[0] = -1; d.f_off[1] = 0; d.f_off[2] = 1;
            d.w_c = 9; d.w_kt = 3; d.w_kf = 1; d.accumulate = 0;
            k.depth.dg = d;
        }
        unit_params(t, k.depth, p + ".depth_conv", p + ".depth_bn", p + ".depth_act");
        k.depth.act = gtt::ACT_PRELU; k.depth.x = k.pc1.a;
        alloc_unit(b, k.depth, nt, 16, lean);
        // point_conv2: Conv2d(16,8,1) weight [8][16] / ConvTranspose2d(16,8,1) weight [16][8]
        k.pc2.cg = conv_geom(B, Tt, Tt, 33, 33, 16, 0, 16, 8, 8, 1, 1, 0, 0, 0, 0, 1, 0, deconv ? 1 : 16, deconv ? 8 : 1, 1, 1);
        unit_params(t, k.pc2, p + ".point_conv2", p + ".point_bn2", "");
        k.pc2.act = gtt::ACT_NONE; k.pc2.x = k.depth.a;
        // point_bn2 has no activation behind it and four readers that can normalise on load (TRALite's energy, the
        // gate/shuffle, and their two backward passes): its output is then never stored (not in the exact-chain mode,
        // whose backward reads 16-bit copies written by that pass)
        k.bn2_load = fuse && !t->exact && (t->fusions & 4096);
        alloc_unit(b, k.pc2, nt, 8, k.bn2_load);
        k.pc2.lean = false;
        k.pc2.deferred = k.bn2_load;
        k.pc2.front = &k.depth;
        k.depth.front = &k.pc1;          // the depth conv's adjoint produces point_conv1's gradient input
        k.pc2.pre = fuse ? &k.depth : nullptr;
        k.depth.deferred = fuse;
        // point_conv1 -> depth_conv: nine global taps each re-applying the BatchNorm + PReLU were slower than the separate
        // pass (312 us against 163 + 107); the LDS-tiled depth convs form the activation once while staging (fusion bit 8)
        const bool fuse_pc1 = fuse && (t->fusions & 256);
        k.depth.pre = fuse_pc1 ? &k.pc1 : nullptr;
        k.pc1.deferred = fuse_pc1;
        k.o_tra = P(t, p + ".tra.depth_conv.weight");
        k.e = b.take((size_t)B * Tt * 8); k.yt = b.take((size_t)B * Tt * 8); k.g = b.take((size_t)B * Tt * 8);
        k.out = (deconv && t->fuse_sums) ? nullptr : b.take_saved(n33 * 16);     // (fused: the next layer's sum buffer)
        k.post = nullptr;
    };
    for (int k = 0; k < 3; ++k) {
        gt_block(t->enc[k], "encoder.en_convs." + std::to_string(k + 2), false, X);
        X = t->enc[k].out;
        skips[k + 2] = X;
        t->taps["en" + std::to_string(k + 2)] = {X, {T, 33, 16}};
    }
    for (int i = 0; i < 8; ++i) {
        TcnBlock& k = t->tcn[i];
        const int d = 1 << (i & 3);
        const std::string p = "gtcn" + std::to_string(i / 4 + 1) + ".blocks." + std::to_string(i & 3);
        k.c1.cg = conv_geom(B, T, T, 33, 33, 16, 0, 16, 16, 16, 1, 1, 0, 0, 0, 0, 1, 0, 16, 1, 1, 1);
        unit_params(t, k.c1, p + ".conv1", p + ".bn1", p + ".act1");
        k.c1.act = gtt::ACT_PRELU; k.c1.x = X;
        alloc_unit(b, k.c1, n33, 16, lean_c1);
        k.c2.dw = true;
        DwGeom g{};
        g.B = B; g.Tin = T; g.Tout = T; g.F = 33; g.C = 16; g.nkt = 3; g.nkf = 1;
        g.t_off[0] = -2 * d; g.t_off[1] = -d; g.t_off[2] = 0; g.f_off[0] = 0;
        g.w_c = 3; g.w_kt = 1; g.w_kf = 1; g.accumulate = 0;
        k.c2.dg = g;
        unit_params(t, k.c2, p + ".conv2", p + ".bn2", p + ".act2");
        k.c2.act = gtt::ACT_PRELU; k.c2.x = k.c1.a;
        alloc_unit(b, k.c2, n33, 16, lean);
        k.c3.cg = k.c1.cg;
        unit_params(t, k.c3, p + ".conv3", p + ".bn3", p + ".act3");
        k.c3.act = gtt::ACT_PRELU; k.c3.x = k.c2.a; k.c3.res = X;
        alloc_unit(b, k.c3, n33, 16, i == 7 && t->fuse_sums);     // (fused: block 7's output lives in dec[0].s)
        k.c3.lean = false;
        k.c3.post = nullptr;
        k.c2.front = &k.c1;
        k.c3.front = &k.c2;
        k.c1.front = i > 0 ? &t->tcn[i - 1].c3 : nullptr;
        k.c2.pre = fuse ? &k.c1 : nullptr; k.c1.deferred = fuse;
        k.c3.pre = fuse ? &k.c2 : nullptr; k.c2.deferred = fuse;
        // conv1 of the NEXT block applies this block's bn3 + residual + PReLU (the last block keeps its own pass)
        k.c1.pre = fuse && i > 0 ? &t->tcn[i - 1].c3 : nullptr;
        if (i > 0) t->tcn[i - 1].c3.deferred = fuse;
        k.c3.deferred = false;
        X = k.c3.a;
        if ((i & 3) == 3) t->taps["gtcn" + std::to_string(i / 4 + 1)] = {X, {T, 33, 16}};
    }
    // bf16 storage (the memory-lean variant): the decoder's sums x + skip (three block inputs, s3, s4) are not saved --
    // one shared buffer holds the current one, the backward recomputes each from its two (saved) addends right before
    // it is needed (five extra streaming adds).  fp32 storage keeps them all, as before.
    t->share_sums = t->bf != 0 && !t->fuse_sums;
    t->s_tmp = t->share_sums ? b.take_saved(n65 * 16) : nullptr;
    t->de3.post = nullptr;
    for (int i = 0; i < 3; ++i) {
        GtBlock& k = t->dec[i];
        k.s = t->share_sums ? t->s_tmp : b.take_saved(n33 * 16);
        k.post = nullptr;
    }
    t->s3 = t->share_sums ? t->s_tmp : (t->fuse_sums ? b.take_saved(n33 * 16) : nullptr);
    if (t->fuse_sums) {
        t->tcn[7].c3.post = t->enc[2].out;              // dec[0].s = gtcn2 output + en_outs[4]
        t->tcn[7].c3.a = t->dec[0].s;
        t->taps["gtcn2"] = {t->dec[0].s, {T, 33, 16}};
        t->tap_minus["gtcn2"] = t->enc[2].out;
    }
    for (int i = 0; i < 3; ++i) {
        GtBlock& k = t->dec[i];
        gt_block(k, "decoder.de_convs." + std::to_string(i), true, k.s);
        k.pc1.x = k.s;
        // the addends of s (previous output, skip) are needed by forward() only
        (void)skips;
        if (t->fuse_sums) {
            // (gt_block took a buffer for the block output: hand the sum's buffer to it instead)
            k.out = i < 2 ? t->dec[i + 1].s : t->s3;
            k.post = i < 2 ? t->enc[1 - i].out : t->en1.a;
            t->tap_minus["de" + std::to_string(i)] = k.post;
        }
        X = k.out;
        t->taps["de" + std::to_string(i)] = {X, {T, 33, 16}};
    }
    if (!t->share_sums && !t->fuse_sums) t->s3 = b.take_saved(n33 * 16);
    // decoder.de_convs.3: ConvTranspose2d(16,16,(1,5),stride (1,2),padding (0,2)), weight [in][out][1][5]
    t->de3.cg = conv_geom(B, T, T, 33, 65, 16, 0, 16, 16, 16, 1, 5, 0, 0, 0, 1, 2, 2, 5, 80, 5, 1);
    unit_params(t, t->de3, "decoder.de_convs.3.conv", "decoder.de_convs.3.bn", "decoder.de_convs.3.act");
    t->de3.act = gtt::ACT_PRELU; t->de3.x = t->s3;
    alloc_unit(b, t->de3, n65, 16, t->fuse_sums);       // (fused: the activation lives in s4 as de3.a + en_outs[0])
    t->de3.lean = false;
    t->s4 = t->share_sums ? t->s_tmp : b.take_saved(n65 * 16);
    if (t->fuse_sums) { t->de3.a = t->s4; t->de3.post = t->en0.a; t->tap_minus["de3"] = t->en0.a; }
    t->de4.cg = conv_geom(B, T, T, 65, 129, 16, 0, 16, 2, 2, 1, 5, 0, 0, 0, 1, 2, 2, 5, 10, 5, 1);
    unit_params(t, t->de4, "decoder.de_convs.4.conv", "decoder.de_convs.4.bn", "");
    t->de4.act = gtt::ACT_TANH; t->de4.x = t->s4;
    alloc_unit(b, t->de4, n129, 2);
    t->taps["de3"] = {t->de3.a, {T, 65, 16}};
    t->taps["de4"] = {t->de4.a, {T, 129, 2}};
    // ---- backward buffers
    const size_t bwd_begin = b.used;
    // (take_grad: the tensors handed from one unit's backward to the next are bf16 in storage mode 5; dy -- a unit's own
    // scratch between its BatchNorm backward and its convs -- and dm / df0 at the two ends of the chain stay fp32)
    t->gs0 = b.take_grad(n65 * 16);
    for (int i = 1; i < 5; ++i) t->gs[i] = b.take_grad(n33 * 16);
    t->q1 = b.take(std::max(b.gsz((size_t)n33 * 16), (size_t)n129 * 3));     // (also dm and df0, fp32: see below)
    t->q2 = b.take_grad(n33 * 16);
    t->dy = b.take((size_t)std::max(std::max(n65 * 16, n33x * 16), n129 * 3));
    // dv, dhd, dh are contiguous: after the last GTConv block they are dead and serve as d65 (gradient of en0.a)
    const size_t sz_dv = (b.gsz((size_t)n33x * 8) + 63) & ~size_t(63), sz_dhd = (b.gsz((size_t)n33x * 16) + 63) & ~size_t(63);
    const size_t pool = sz_dv + sz_dhd + b.gsz((size_t)n33 * 16);
    float* pl = b.take(std::max(pool + 192, b.gsz((size_t)n65 * 16)));
    t->dv = pl;
    t->dhd = pl ? pl + sz_dv : nullptr;
    t->dh = pl ? t->dhd + sz_dhd : nullptr;
    t->d65 = pl;
    t->tmp_tra = b.take((size_t)B * T2 * 8 * 3);
    // dm (gradient of the mask, consumed by the first backward unit) and df0 (gradient of the SFE output, produced by
    // the last one) live in q1, which the TCN / encoder backward uses only in between
    static_assert(129 * 3 <= 33 * 16 && 129 * 2 <= 33 * 16, "dm / df0 must fit in q1");
    t->dm = t->q1;
    t->df0 = t->q1;
    if (!t->exact) return b.used;
    // ---- exact chain: the fp32 twins, in forward order, each alive from its producer to its last forward reader
    ChainPool cp;
    float* cbase = base ? base + bwd_begin : nullptr;
    auto get = [&](size_t n) -> float* {
        const size_t o = cp.get(n);
        return cbase ? cbase + o : reinterpret_cast<float*>(sizeof(float) * (o + 64));     // (size query: a fake, distinct address)
    };
    auto put = [&](const float* p, size_t n) {
        const float* b0 = cbase ? cbase : reinterpret_cast<const float*>(sizeof(float) * 64);
        cp.put((size_t)(p - b0), n);
    };
    const size_t N129 = (size_t)n129 * 3, N65 = (size_t)n65 * 16, N33 = (size_t)n33 * 16;
    t->ebc = get(N129);
    t->f0c = get(N129);
    put(t->ebc, N129);                                              // feat -> SFE
    auto plain_unit = [&](Unit& u, const float* xin, size_t ny, size_t nxin, bool free_in) {
        u.xc = xin;
        u.yc = get(ny);
        if (free_in) put(xin, nxin);
        u.ac = get(ny);
        put(u.yc, ny);
    };
    plain_unit(t->en0, t->f0c, N65, N129, true);                    // en0.ac lives until s4
    plain_unit(t->en1, t->en0.ac, N33, N65, false);                 // en1.ac lives until s3
    auto gt_chain = [&](GtBlock& k, const float* xin, bool deconv) {
        const size_t nt16 = (size_t)(deconv ? n33x : n33) * 16, nt8 = nt16 / 2;
        k.xinc = xin;
        k.pc1.xc = xin;
        k.pc1.yc = get(N33);
        if (!k.pc1.deferred) { k.pc1.ac = get(N33); put(k.pc1.yc, N33); } else k.pc1.ac = nullptr;
        k.depth.xc = k.pc1.deferred ? nullptr : k.pc1.ac;
        k.depth.yc = get(nt16);
        put(k.pc1.deferred ? k.pc1.yc : k.pc1.ac, N33);
        if (!k.depth.deferred) { k.depth.ac = get(nt16); put(k.depth.yc, nt16); } else k.depth.ac = nullptr;
        k.pc2.xc = k.depth.deferred ? nullptr : k.depth.ac;
        k.pc2.yc = get(nt8);
        if (k.depth.deferred) put(k.depth.yc, nt16); else put(k.depth.ac, nt16);
        k.pc2.ac = get(nt8);
        put(k.pc2.yc, nt8);
        k.outc = get(N33);
        put(k.pc2.ac, nt8);
    };
    const float* Xc = t->en1.ac;
    for (int k = 0; k < 3; ++k) {                                   // the encoder outputs live until the decoder's adds
        gt_chain(t->enc[k], Xc, false);
        Xc = t->enc[k].outc;
    }
    {
        const float* Xin = Xc;                                      // fp32 input of the block being laid out
        for (int i = 0; i < 8; ++i) {
            TcnBlock& k = t->tcn[i];
            if (k.c1.pre) {
                // conv1 finishes the previous block: it reads that block's conv3 output and residual input and writes
                // its activation -- this block's input and residual -- in both forms
                Unit& pv = t->tcn[i - 1].c3;
                pv.ac = get(N33);
                k.c1.yc = get(N33);
                put(pv.yc, N33);
                if (i - 1 > 0) put(pv.resc, N33);                   // (block 0's input is enc[2].outc: a skip, it stays)
                Xin = pv.ac;
                k.c1.xc = nullptr;
            } else {
                k.c1.xc = Xin;
                k.c1.yc = get(N33);
            }
            k.c3.resc = Xin;
            auto next_unit = [&](Unit& u, Unit& v) {                // u reads v: fused v.yc, else v.ac
                if (!v.deferred) { v.ac = get(N33); put(v.yc, N33); }
                u.xc = v.deferred ? nullptr : v.ac;
                u.yc = get(N33);
                put(v.deferred ? v.yc : v.ac, N33);
            };
            next_unit(k.c2, k.c1);
            next_unit(k.c3, k.c2);
            if (!k.c3.deferred) {                                   // its own bn_act pass (the last block; or no fusion)
                k.c3.ac = get(N33);
                put(k.c3.yc, N33);
                if (i > 0) put(Xin, N33);
                Xin = k.c3.ac;
            }
        }
        Xc = Xin;
    }
    for (int i = 0; i < 3; ++i) {
        GtBlock& k = t->dec[i];
        t->dec_in[i] = Xc;
        k.sc = get(N33);
        put(Xc, N33);
        put(t->enc[2 - i].outc, N33);
        gt_chain(k, k.sc, true);
        put(k.sc, N33);
        Xc = k.outc;
    }
    t->s3c = get(N33);
    put(Xc, N33);
    put(t->en1.ac, N33);
    plain_unit(t->de3, t->s3c, N65, N33, true);
    t->s4c = get(N65);
    put(t->de3.ac, N65);
    put(t->en0.ac, N65);
    plain_unit(t->de4, t->s4c, (size_t)n129 * 2, N65, true);
    put(t->de4.ac, (size_t)n129 * 2);
    return std::max(b.used, bwd_begin + cp.hi);
}

int ensure_plan(gtcrn_trainer* t, int B, int T) {
    if (t->planned && t->B == B && t->T == T) return 0;
    const size_t need = plan(t, B, T, nullptr);
    if (need > t->arena_floats) {
        T_HIP(hipDeviceSynchronize());
        if (t->arena) (void)hipFree(t->arena);
        t->arena = nullptr; t->arena_floats = 0;
        T_HIP(hipMalloc(&t->arena, need * sizeof(float)));
        t->arena_floats = need;
    }
    plan(t, B, T, t->arena);
    t->B = B; t->T = T; t->planned = true; t->have_fwd = false;
    t->shift_ready = false;       // new arena: the first forward seeds the shifts from the running means
    return 0;
}

int unit_fwd(gtcrn_trainer* t, Unit& u, float* prm, hipStream_t s) {
    int parts = 0;   // > 0: the conv kernel produced the BatchNorm partial sums in its epilogue
    float* bn = prm + u.o_bn;   // weight, bias, running_mean, running_var (consecutive in the blob)
    // bf16 storage: y is stored centred on the channel's batch mean of the previous step (the running mean before the
    // first one); the train-mode BatchNorm is shift invariant and bn_stats keeps the shift up to date
    float* shift = (t->ybf && !t->exact) ? u.stats + 32 : nullptr;
    if (shift && !t->shift_ready)
        T_HIP(hipMemcpyAsync(shift, bn + 2 * u.C, sizeof(float) * u.C, hipMemcpyDeviceToDevice, s));
    if (t->exact) {
        // bf16 saves, exact chain: the unit runs exactly as in fp32 storage on the chain's fp32 tensors; the 16-bit
        // copies for the backward are second stores -- of the activation by whoever produces it, of the conv output y by
        // whoever CONSUMES it (this unit's bn_act, or the next unit's normalise-on-load conv): centred on this step's
        // mean and kept on the forward's side of the PReLU kink (gtt::BnPre::y_out, bn_act's y2)
        // (the statistics' second stage runs in the conv kernel's last workgroup when fusion bit 10 is on: parts < 0)
        const gtt::StatFin sf{u.n, u.C, u.stats, bn + 2 * u.C, bn + 3 * u.C, nullptr, u.stats + 64};
        if (u.pre) {
            const Unit& v = *u.pre;
            const float* vbn = prm + v.o_bn;
            gtt::BnPre bp{};
            bp.stats = v.stats; bp.gamma = vbn; bp.beta = vbn + v.C;
            bp.slope = v.o_slope >= 0 ? prm + v.o_slope : nullptr;
            bp.a_out = v.a; bp.ybf = 0; bp.bf = t->bf;
            bp.res = v.resc;
            bp.exact = 1; bp.a_chain = v.ac; bp.y_out = v.y; bp.ybf_out = t->ybf;
            if (u.dw) {
                DwGeom g = u.dg;
                g.in_bf = 0; g.out_bf = 0;
                T_RUN(gtt::dw_fwd(g, v.yc, prm + u.o_w, prm + u.o_b, u.yc, s, t->dscratch, &parts, nullptr, &bp, nullptr, 0, &sf));
            } else {
                ConvGeom g = u.cg;
                g.in_bf = 0; g.out_bf = 0;
                T_RUN(gtt::conv_fwd(g, v.yc, prm + u.o_w, prm + u.o_b, u.yc, s, t->dscratch, &parts, nullptr, &bp, nullptr, 0, &sf));
            }
        } else if (u.dw) {
            DwGeom g = u.dg;
            g.in_bf = 0; g.out_bf = 0;
            T_RUN(gtt::dw_fwd(g, u.xc, prm + u.o_w, prm + u.o_b, u.yc, s, t->dscratch, &parts, nullptr, nullptr, nullptr, 0, &sf));
        } else {
            ConvGeom g = u.cg;
            g.in_bf = 0; g.out_bf = 0;
            T_RUN(gtt::conv_fwd(g, u.xc, prm + u.o_w, prm + u.o_b, u.yc, s, t->dscratch, &parts, nullptr, nullptr, nullptr, 0, &sf));
        }
        T_RUN(gtt::bn_stats(u.yc, u.n, u.C, u.stats, bn + 2 * u.C, bn + 3 * u.C, t->dscratch, s, parts, 0, nullptr,
                            u.stats + 64));
        if (!u.deferred)
            T_RUN(gtt::bn_act(u.yc, u.n, u.C, u.stats, bn, bn + u.C, u.resc, u.act, u.o_slope >= 0 ? prm + u.o_slope : nullptr,
                              u.ac, s, 0, 0, u.a, t->bf, u.y, t->ybf));
        return 0;
    }
    const gtt::StatFin sf{u.n, u.C, u.stats, bn + 2 * u.C, bn + 3 * u.C, shift, nullptr};
    if (u.pre) {
        const Unit& v = *u.pre;
        const float* vbn = prm + v.o_bn;
        gtt::BnPre bp{};
        bp.stats = v.stats; bp.gamma = vbn; bp.beta = vbn + v.C;
        bp.slope = v.o_slope >= 0 ? prm + v.o_slope : nullptr;
        bp.a_out = v.a; bp.ybf = t->ybf; bp.bf = t->bf;
        bp.res = v.res;
        if (u.dw) {
            DwGeom g = u.dg;
            g.in_bf = t->ybf;
            T_RUN(gtt::dw_fwd(g, v.y, prm + u.o_w, prm + u.o_b, u.y, s, t->dscratch, &parts, shift, &bp, nullptr, 0, &sf));
        } else {
            ConvGeom g = u.cg;
            g.in_bf = t->ybf;
            T_RUN(gtt::conv_fwd(g, v.y, prm + u.o_w, prm + u.o_b, u.y, s, t->dscratch, &parts, shift, &bp, nullptr, 0, &sf));
        }
    }
    else if (u.dw) T_RUN(gtt::dw_fwd(u.dg, u.x, prm + u.o_w, prm + u.o_b, u.y, s, t->dscratch, &parts, shift, nullptr, nullptr, 0, &sf));
    else T_RUN(gtt::conv_fwd(u.cg, u.x, prm + u.o_w, prm + u.o_b, u.y, s, t->dscratch, &parts, shift, nullptr, nullptr, 0, &sf));
    T_RUN(gtt::bn_stats(u.y, u.n, u.C, u.stats, bn + 2 * u.C, bn + 3 * u.C, t->dscratch, s, parts, t->ybf, shift));
    if (!u.deferred)
        T_RUN(gtt::bn_act(u.y, u.n, u.C, u.stats, bn, bn + u.C, u.res, u.act, u.o_slope >= 0 ? prm + u.o_slope : nullptr,
                          u.a, s, t->bf, t->ybf, nullptr, 0, nullptr, 0, u.post));
    return 0;
}

// da: gradient w.r.t. u.a (read only).  dx: where the data gradient goes (nullptr: not needed).
// da_bf / dx_bf (-1: the trainer's gradient format t->gbf): storage format of da (and dres) / of dx.  Everything between two
// units is in the trainer's format; the two ends of the chain -- dm, the mask's gradient, and df0, the SFE output's -- are fp32.
int unit_bwd(gtcrn_trainer* t, Unit& u, const float* prm, float* grads, const float* da, float* dx, int dx_acc,
             float* dres, int dres_acc, hipStream_t s, int da_bf = -1, int dx_bf = -1) {
    if (da_bf < 0) da_bf = t->gbf;
    if (dx_bf < 0) dx_bf = t->gbf;
    const int gb = (da_bf && dx_bf) ? 1 : 0;          // the fused backward kernels take ONE gradient format
    const bool fused_ok = da_bf == dx_bf;             // (a unit whose two sides differ takes the pass-by-pass form)
    const float* bn = prm + u.o_bn;
    float* gbn = grads + u.o_bn;
    // the BatchNorm reduction of this unit may already sit in dscratch (left by the kernel that produced da)
    const int have_parts = t->red_unit == &u ? t->red_parts : 0;
    t->red_unit = nullptr;
    // the unit in front (backward order) whose gradient input is this unit's dx: its reduction can ride along
    const Unit* f = u.front;
    const bool ride = f && dx && f->C == 16 && f->act == gtt::ACT_PRELU && f->o_slope >= 0 && t->bf == t->ybf && t->bf <= 1 &&
                      (t->fusions & 4);
    gtt::DwUnitNext nx{};
    if (ride) {
        const float* fbn = prm + f->o_bn;
        nx.y = f->y; nx.stats = f->bstats; nx.gamma = fbn; nx.beta = fbn + f->C; nx.slope = prm + f->o_slope;
        nx.res = f->res;
        // (in-launch finish, fusion bit 10: where that unit's parameter gradients go; the launchers ignore it when off)
        nx.dgamma = grads + f->o_bn; nx.dbeta = grads + f->o_bn + f->C; nx.dslope = grads + f->o_slope; nx.n = f->n;
        // this unit's input IS that unit's activation (conv3 <- conv2 <- conv1 <- the previous block's conv3,
        // point_conv2 <- depth_conv): recomputed from the y the reduction reads anyway instead of loaded
        if ((t->fusions & 8) && u.x == f->a)
            nx.recompute_x = (t->bf && !t->exact) ? 2 : 1;
    }
    if (fused_ok && !u.dw && u.cg.nkt == 1 && u.cg.nkf == 1 && u.cg.sf == 1 && (u.cg.Cin % 4) == 0 && (u.cg.Cout % 4) == 0) {
        // pointwise unit: BatchNorm backward, data gradient and weight gradient in one pass (after the reduction)
        int parts = 0;
        T_RUN(gtt::unit1x1_bwd(u.cg, u.x, u.y, da, u.res, u.bstats, bn, bn + u.C, u.act,
                               u.o_slope >= 0 ? prm + u.o_slope : nullptr, prm + u.o_w, dx, dx_acc, dres, dres_acc,
                               grads + u.o_w, u.o_b >= 0 ? grads + u.o_b : nullptr, gbn, gbn + u.C,
                               u.o_slope >= 0 ? grads + u.o_slope : nullptr, t->dscratch, t->fscratch, s, t->bf,
                               t->ybf, have_parts, ride && f->n == u.n ? &nx : nullptr, &parts, gb));
        if (parts != 0) { t->red_unit = f; t->red_parts = parts; }
        return 0;
    }
    if (fused_ok && u.dw && u.C == 16 && u.dg.nkt == 3 && u.dg.nkf == 1 && u.act == gtt::ACT_PRELU && !u.res && u.o_slope >= 0 &&
        dx && !dx_acc && !dres && t->bf == t->ybf && t->bf <= 1 && (t->fusions & 2)) {
        // TCN conv2: dy, weight gradient and data gradient in one pass; conv1's reduction rides along
        const bool ride2 = ride && !f->res && f->n == u.n;
        int parts = 0;
        T_RUN(gtt::dwunit_bwd(u.dg, u.x, u.y, da, u.bstats, bn, bn + u.C, prm + u.o_slope, prm + u.o_w, dx,
                              grads + u.o_w, u.o_b >= 0 ? grads + u.o_b : nullptr, gbn, gbn + u.C, grads + u.o_slope,
                              t->dscratch, t->fscratch, s, t->bf, t->ybf, ride2 ? &nx : nullptr, &parts, have_parts, gb));
        if (ride2 && parts != 0) { t->red_unit = f; t->red_parts = parts; }
        return 0;
    }
    if (fused_ok && u.dw && u.C == 16 && u.dg.nkt == 3 && u.dg.nkf == 3 && u.dg.F == 33 && u.dg.Tin == u.dg.Tout && u.act == gtt::ACT_PRELU &&
        !u.res && u.o_slope >= 0 && dx && !dx_acc && !dres && t->bf == t->ybf && t->bf <= 1 && (t->fusions & 64)) {
        // encoder depth_conv: dy, weight gradient and data gradient in one LDS-tiled pass; point_conv1's reduction rides along
        const bool ride3 = ride && !f->res && f->n == u.n;
        int parts = 0;
        T_RUN(gtt::dwunit33_bwd(u.dg, u.x, u.y, da, u.bstats, bn, bn + u.C, prm + u.o_slope, prm + u.o_w, dx,
                                grads + u.o_w, u.o_b >= 0 ? grads + u.o_b : nullptr, gbn, gbn + u.C, grads + u.o_slope,
                                t->dscratch, t->fscratch, s, t->bf, t->ybf, ride3 ? &nx : nullptr, &parts, have_parts, gb));
        if (ride3 && parts != 0) { t->red_unit = f; t->red_parts = parts; }
        return 0;
    }
    if (fused_ok && !u.dw && u.cg.nkt == 3 && u.cg.nkf == 3 && u.cg.f_mode == 1 && u.cg.Tout == u.cg.Tin + 2 && u.C == 16 &&
        u.act == gtt::ACT_PRELU && !u.res && u.o_slope >= 0 && dx && !dx_acc && !dres && t->bf == t->ybf && t->bf <= 1 &&
        (t->fusions & 128)) {
        // decoder depth_conv (dense transposed 3x3): dy and both matrix products from LDS tiles; point_conv1's reduction rides
        const bool ride3 = ride && !f->res && f->n == (long)u.cg.B * u.cg.Tin * u.cg.Fin;
        int parts = 0;
        T_RUN(gtt::dense33_bwd(u.cg, u.x, u.y, da, u.bstats, bn, bn + u.C, prm + u.o_slope, prm + u.o_w, dx, grads + u.o_w,
                               u.o_b >= 0 ? grads + u.o_b : nullptr, gbn, gbn + u.C, grads + u.o_slope, t->dscratch,
                               t->fscratch, s, t->bf, t->ybf, ride3 ? &nx : nullptr, &parts, have_parts, gb));
        if (ride3 && parts != 0) { t->red_unit = f; t->red_parts = parts; }
        return 0;
    }
    if (fused_ok && !u.dw && u.cg.nkt == 1 && u.cg.nkf == 5 && u.cg.sf == 2 && u.cg.Cin == 16 && u.cg.Cout == 16 && u.C == 16 &&
        u.act == gtt::ACT_PRELU && !u.res && u.o_slope >= 0 && dx && !dres && u.x && t->bf == t->ybf && t->bf <= 1 &&
        (t->fusions & 512)) {
        // en_convs.1 / de_convs.3: dy and both matrix products from LDS tiles; en_convs.0's reduction rides on en_convs.1's dx
        const bool ride5 = ride && (t->fusions & 32) && !f->res && u.cg.f_mode == 0 && f->n == (long)u.cg.B * u.cg.Tin * u.cg.Fin;
        int parts = 0;
        T_RUN(gtt::conv15_bwd(u.cg, u.x, u.y, da, u.bstats, bn, bn + u.C, prm + u.o_slope, prm + u.o_w, dx, dx_acc,
                              grads + u.o_w, u.o_b >= 0 ? grads + u.o_b : nullptr, gbn, gbn + u.C, grads + u.o_slope,
                              t->dscratch, t->fscratch, s, t->bf, t->ybf, ride5 ? &nx : nullptr, &parts, have_parts, gb));
        if (ride5 && parts != 0) { t->red_unit = f; t->red_parts = parts; }
        return 0;
    }
    T_RUN(gtt::bn_act_bwd(da, u.y, u.n, u.C, u.bstats, bn, bn + u.C, u.res, u.act,
                          u.o_slope >= 0 ? prm + u.o_slope : nullptr, t->dy, dres, dres_acc, gbn, gbn + u.C,
                          u.o_slope >= 0 ? grads + u.o_slope : nullptr, t->dscratch, s, t->bf, t->ybf, have_parts, da_bf));
    // the adjoint conv produces the gradient input of the unit in front: that unit's reduction rides in its epilogue
    // (depthwise / dense 3x3 -> point_conv1, en_convs.1 -> en_convs.0; fusion bit 5).  dscratch is free again: this
    // unit's own partial sums were consumed by bn_act_bwd above
    bool ride_adj = ride && (t->fusions & 32) && !f->res && dx && !u.res && !dres;
    if (ride_adj) {
        if (u.dw) ride_adj = u.C == 16 && u.dg.nkt == 3 && u.dg.nkf == 3 && f->n == (long)u.dg.B * u.dg.Tin * u.dg.F;
        else ride_adj = u.cg.Cin == 16 && u.cg.CinT == 16 && u.cg.cin_off == 0 && (u.cg.Cin % 4) == 0 && (u.cg.Cout % 4) == 0 &&
                        ((u.cg.nkt == 3 && u.cg.nkf == 3) || (u.cg.nkt == 1 && u.cg.nkf == 5)) &&
                        f->n == (long)u.cg.B * u.cg.Tin * u.cg.Fin;
    }
    int parts = 0;
    if (u.dw) {
        T_RUN(gtt::dw_wgrad(u.dg, u.x, t->dy, grads + u.o_w, u.o_b >= 0 ? grads + u.o_b : nullptr, t->fscratch, s));
        DwGeom ag = adjoint(u.dg, dx_acc);
        ag.out_bf = dx_bf;                            // (dy, the adjoint's input, is this unit's fp32 scratch)
        if (dx) T_RUN(gtt::dw_fwd(ag, t->dy, prm + u.o_w, nullptr, dx, s, ride_adj ? t->dscratch : nullptr,
                                  ride_adj ? &parts : nullptr, nullptr, nullptr, ride_adj ? &nx : nullptr, t->ybf));
    } else {
        T_RUN(gtt::conv_wgrad(u.cg, u.x, t->dy, grads + u.o_w, u.o_b >= 0 ? grads + u.o_b : nullptr, t->fscratch, s));
        ConvGeom ag = adjoint(u.cg, dx_acc);
        ag.out_bf = dx_bf;
        if (dx) T_RUN(gtt::conv_fwd(ag, t->dy, prm + u.o_w, nullptr, dx, s, ride_adj ? t->dscratch : nullptr,
                                    ride_adj ? &parts : nullptr, nullptr, nullptr, ride_adj ? &nx : nullptr, t->ybf));
    }
    if (ride_adj && parts != 0) { t->red_unit = f; t->red_parts = parts; }
    return 0;
}

int gt_fwd(gtcrn_trainer* t, GtBlock& k, float* prm, hipStream_t s) {
    int rc;
    if ((rc = unit_fwd(t, k.pc1, prm, s))) return rc;
    if ((rc = unit_fwd(t, k.depth, prm, s))) return rc;
    if ((rc = unit_fwd(t, k.pc2, prm, s))) return rc;
    const float* tr = prm + k.o_tra;   // depth_conv.weight[24], .bias[8], point_conv.weight[64], .bias[8]
    if (t->exact) {
        T_RUN(gtt::tra_fwd(k.pc2.ac, t->B, k.Tt, tr, tr + 24, tr + 32, tr + 96, k.e, k.yt, k.g, s, 0));
        T_RUN(gtt::gate_shuffle_fwd(k.pc2.ac, k.g, k.xinc, t->B, t->T, k.Tt, k.outc, s, 0, k.out, t->bf));
        return 0;
    }
    const float* bn2 = prm + k.pc2.o_bn;
    const gtt::TraBn tb{k.pc2.stats, bn2, bn2 + 8, t->ybf};
    const gtt::TraBn* ptb = k.bn2_load ? &tb : nullptr;
    const float* v = k.bn2_load ? k.pc2.y : k.pc2.a;
    T_RUN(gtt::tra_fwd(v, t->B, k.Tt, tr, tr + 24, tr + 32, tr + 96, k.e, k.yt, k.g, s, t->bf, ptb));
    T_RUN(gtt::gate_shuffle_fwd(v, k.g, k.xin, t->B, t->T, k.Tt, k.out, s, t->bf, nullptr, 0, k.post, ptb));
    return 0;
}
// dout: gradient of k.out; dxin: gradient of the block input (all 16 channels written; acc: ADDED to what dxin holds --
// the gradient the same tensor received as a decoder skip)
int gt_bwd(gtcrn_trainer* t, GtBlock& k, const float* prm, float* grads, const float* dout, float* dxin, int acc,
           hipStream_t s) {
    const float* tr = prm + k.o_tra;
    float* gtr = grads + k.o_tra;
    int rc;
    const float* bn2 = prm + k.pc2.o_bn;
    const gtt::TraBn tb{k.pc2.bstats, bn2, bn2 + 8, t->ybf};
    T_RUN(gtt::tra_gate_shuffle_bwd(dout, k.bn2_load ? k.pc2.y : k.pc2.a, k.g, k.e, k.yt, t->B, t->T, k.Tt, tr, tr + 32, t->dv,
                                    dxin, gtr, gtr + 24, gtr + 32, gtr + 96, t->tmp_tra, t->fscratch, s, t->bf, acc,
                                    k.bn2_load ? &tb : nullptr, t->gbf));
    if ((rc = unit_bwd(t, k.pc2, prm, grads, t->dv, t->dhd, 0, nullptr, 0, s))) return rc;
    if ((rc = unit_bwd(t, k.depth, prm, grads, t->dhd, t->dh, 0, nullptr, 0, s))) return rc;
    if ((rc = unit_bwd(t, k.pc1, prm, grads, t->dh, dxin, acc, nullptr, 0, s))) return rc;   // channels 0..7
    return 0;
}

}  // namespace

extern "C" {

int gtcrn_trainer_create(gtcrn_trainer** out, int device) {
    if (!out) return tfail(GTCRN_ERR_ARG, "gtcrn_trainer_create: null out");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return tfail(GTCRN_ERR_DEVICE, "gtcrn_trainer_create: no such HIP device (the training path has no CPU fallback)");
    hipDeviceProp_t prop;
    T_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return tfail(GTCRN_ERR_DEVICE, std::string("gtcrn_trainer_create: built for gfx950 only, device is ") + prop.gcnArchName);
    T_HIP(hipSetDevice(device));
    gtcrn_trainer* t = new gtcrn_trainer();
    t->device = device;
    for (const auto& p : gtcrn::param_table()) t->off[p.name] = p.offset;
    hipError_t e = hipMalloc(&t->fscratch, sizeof(float) * gtt::MAX_PARTIALS * (9 * 256 + 16));
    if (e == hipSuccess) e = hipMalloc(&t->dscratch, sizeof(double) * gtt::MAX_PARTIALS * 3 * 16 + 256);
    // (the 64 MiB pool of the batched weight-gradient finishes, fusion bit 15, is allocated by the first backward that
    // uses it -- ensure_wfin_pool -- and counted by the workspace queries)
    if (e == hipSuccess) e = hipMalloc(&t->fin_gpart, sizeof(double) * gtt::FIN_GPART_DOUBLES);
    if (e == hipSuccess) e = hipMalloc(&t->fin_ctr, sizeof(unsigned) * gtt::FIN_CTR_WORDS);
    if (e == hipSuccess) e = hipMemset(t->fin_ctr, 0, sizeof(unsigned) * gtt::FIN_CTR_WORDS);
    if (e != hipSuccess) {
        gtcrn_trainer_destroy(t);
        return tfail(GTCRN_ERR_HIP, std::string("gtcrn_trainer_create: hipMalloc: ") + hipGetErrorString(e));
    }
    *out = t;
    return 0;
}

void gtcrn_trainer_destroy(gtcrn_trainer* t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->arena) (void)hipFree(t->arena);
    if (t->fscratch) (void)hipFree(t->fscratch);
    if (t->wfin_pool) (void)hipFree(t->wfin_pool);
    if (t->dscratch) (void)hipFree(t->dscratch);
    if (t->fin_gpart) (void)hipFree(t->fin_gpart);
    if (t->fin_ctr) (void)hipFree(t->fin_ctr);
    if (t->loss_ws) (void)hipFree(t->loss_ws);
    if (t->d_win) (void)hipFree(t->d_win);
    delete t;
}

static int storage_formats(int storage, int* bf, int* ybf) {
    // 0: fp32 | 1: bf16 | 4: bf16 SAVES only -- what the backward re-reads is stored as in mode 1, but the forward chain
    // itself stays fp32 (every forward tensor is written twice): the forward IS the fp32 network's, the gradient differs
    // from it only by the rounding of the saved tensors | 5: mode 1 with the gradients handed between units in bf16 too
    // (round 5: what bf16 autocast training stores; the forward is mode 1's bit for bit).
    // (2, 3 were diagnostics of rounds 2-4 -- bf16 activations with fp32 / centred-fp16 conv outputs, used to find out where
    // the bf16 gradient noise comes from (neither lowers it: it enters through the activations).  Removed in round 5 with
    // the fp16 format they needed: its run-time format switch cost every 16-bit conversion of the bf16 modes.)
    if (storage != 0 && storage != 1 && storage != 4 && storage != 5) return -1;
    *bf = storage ? 1 : 0;
    *ybf = storage ? 1 : 0;
    return 0;
}

// what a trainer holds besides the planned arena: the partial-sum scratch of the weight-gradient kernels and, with fusion
// bit 15, the pool their partials wait in for the two batched finish launches
static size_t fixed_workspace_bytes(int fusions) {
    return sizeof(float) * gtt::MAX_PARTIALS * (9 * 256 + 16) + sizeof(double) * gtt::MAX_PARTIALS * 3 * 16 + 256 +
           sizeof(double) * gtt::FIN_GPART_DOUBLES + sizeof(unsigned) * gtt::FIN_CTR_WORDS +
           ((fusions & 32768) ? sizeof(float) * WFIN_POOL_FLOATS : 0);
}
static int ensure_wfin_pool(gtcrn_trainer* t) {
    if (!(t->fusions & 32768) || t->wfin_pool) return 0;
    T_HIP(hipMalloc(&t->wfin_pool, sizeof(float) * WFIN_POOL_FLOATS));
    return 0;
}

long gtcrn_train_workspace_bytes2(int B, int T, int storage) {
    gtcrn_trainer tmp;
    if (storage_formats(storage, &tmp.bf, &tmp.ybf)) return -1;
    tmp.exact = storage == 4;
    tmp.gbf = storage == 5;
    for (const auto& p : gtcrn::param_table()) tmp.off[p.name] = p.offset;
    return (long)(plan(&tmp, B, T, nullptr) * sizeof(float) + fixed_workspace_bytes(tmp.fusions));
}
long gtcrn_train_workspace_bytes(int B, int T) { return gtcrn_train_workspace_bytes2(B, T, 0); }

long gtcrn_trainer_workspace_bytes(gtcrn_trainer* t, int B, int T) {
    if (!t || B < 1 || T < 1) return -1;
    gtcrn_trainer tmp;                 // planned with THIS trainer's storage mode and fusion mask (a mask that stores
    tmp.bf = t->bf;                    // every activation needs about 6 GiB more at B = 512 than the default)
    tmp.ybf = t->ybf;
    tmp.exact = t->exact;
    tmp.gbf = t->gbf;
    tmp.fusions = t->fusions;
    tmp.off = t->off;
    return (long)(plan(&tmp, B, T, nullptr) * sizeof(float) + fixed_workspace_bytes(t->fusions));
}

int gtcrn_trainer_set_storage(gtcrn_trainer* t, int storage) {
    int bf = 0, ybf = 0;
    if (!t || storage_formats(storage, &bf, &ybf))
        return tfail(GTCRN_ERR_ARG, "gtcrn_trainer_set_storage: storage must be 0 (fp32), 1 (bf16), 4 (bf16 saves, fp32 "
                                    "forward chain) or 5 (bf16 with bf16 gradient hand-offs)");
    const int exact = storage == 4, gbf = storage == 5;
    if (t->bf != bf || t->ybf != ybf || t->exact != exact || t->gbf != gbf) {
        t->bf = bf;
        t->ybf = ybf;
        t->exact = exact;
        t->gbf = gbf;
        t->planned = false;      // the arena is re-laid out on the next forward
        t->have_fwd = false;
    }
    return 0;
}

int gtcrn_trainer_set_fusions(gtcrn_trainer* t, int mask) {
    if (!t || mask < 0 || mask > 65535) return tfail(GTCRN_ERR_ARG, "gtcrn_trainer_set_fusions: mask must be 0..65535");
    if (t->fusions != mask) {
        t->fusions = mask;
        t->planned = false;      // the unit links are laid out again on the next forward
        t->have_fwd = false;
    }
    return 0;
}

int gtcrn_train_forward(gtcrn_trainer* t, float* d_params, const float* d_spec, long sb, long sf, long st,
                        float* d_out, long ob, long of, long ot, int B, int T, void* stream) {
    if (!t || !d_params || !d_spec || !d_out || B < 1 || T < 1)
        return tfail(GTCRN_ERR_ARG, "gtcrn_train_forward: bad argument");
    T_HIP(hipSetDevice(t->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_plan(t, B, T);
    if (rc) return rc;
    // in-launch finish of the BatchNorm reductions: this call's context; the arrival counters start at zero whatever a
    // failed or interrupted call may have left in them
    gtt::set_fin_context((t->fusions & 1024) != 0, t->fin_gpart, t->fin_ctr);
    gtt::set_pointwise_form((t->fusions & 8192) != 0);
    gtt::set_column_form((t->fusions & 16384) != 0);
    T_HIP(hipMemsetAsync(t->fin_ctr, 0, sizeof(unsigned) * gtt::FIN_CTR_WORDS, s));
    float* prm = d_params;
    const bool ex = t->exact != 0;
    if (ex) T_RUN(gtt::feat_fwd(d_spec, sb, sf, st, B, T, prm + P(t, "erb.erb_fc.weight"), t->ebc, s, 0, t->eb, t->bf));
    else T_RUN(gtt::feat_fwd(d_spec, sb, sf, st, B, T, prm + P(t, "erb.erb_fc.weight"), t->eb, s, t->bf));
    {   // SFE_Lite: Conv2d(3,3,(1,3),padding (0,1),groups 3,bias=False), weight [3][1][1][3]
        DwGeom g{};
        g.B = B; g.Tin = T; g.Tout = T; g.F = 129; g.C = 3; g.nkt = 1; g.nkf = 3;
        g.in_bf = g.out_bf = t->bf;
        g.f_off[0] = -1; g.f_off[1] = 0; g.f_off[2] = 1; g.w_c = 3; g.w_kt = 3; g.w_kf = 1;
        if (ex) {
            g.in_bf = g.out_bf = 0; g.out2 = t->f0; g.out2_bf = t->bf;
            T_RUN(gtt::dw_fwd(g, t->ebc, prm + P(t, "sfe.depth_conv.weight"), nullptr, t->f0c, s));
        } else {
            T_RUN(gtt::dw_fwd(g, t->eb, prm + P(t, "sfe.depth_conv.weight"), nullptr, t->f0, s));
        }
    }
    if ((rc = unit_fwd(t, t->en0, prm, s))) return rc;
    if ((rc = unit_fwd(t, t->en1, prm, s))) return rc;
    for (int k = 0; k < 3; ++k)
        if ((rc = gt_fwd(t, t->enc[k], prm, s))) return rc;
    for (int i = 0; i < 8; ++i) {
        if ((rc = unit_fwd(t, t->tcn[i].c1, prm, s))) return rc;
        if ((rc = unit_fwd(t, t->tcn[i].c2, prm, s))) return rc;
        if ((rc = unit_fwd(t, t->tcn[i].c3, prm, s))) return rc;
    }
    const long n33 = (long)B * T * 33 * 16, n65 = (long)B * T * 65 * 16;
    if (ex) {
        // the decoder's sums x + skip in fp32 (the backward recomputes its own from the 16-bit copies, see plan())
        for (int i = 0; i < 3; ++i) {
            T_RUN(gtt::add(t->dec_in[i], t->enc[2 - i].outc, t->dec[i].sc, n33, s));
            if ((rc = gt_fwd(t, t->dec[i], prm, s))) return rc;
        }
        T_RUN(gtt::add(t->dec[2].outc, t->en1.ac, t->s3c, n33, s));
        if ((rc = unit_fwd(t, t->de3, prm, s))) return rc;
        T_RUN(gtt::add(t->de3.ac, t->en0.ac, t->s4c, n65, s));
        if ((rc = unit_fwd(t, t->de4, prm, s))) return rc;
        T_RUN(gtt::bs_mask_fwd(t->de4.ac, d_spec, sb, sf, st, B, T, prm + P(t, "erb.ierb_fc.weight"), d_out, ob, of, ot, s, 0));
        t->have_fwd = true;
        t->shift_ready = true;
        return 0;
    }
    const float* X = t->tcn[7].c3.a;
    const bool fs = t->fuse_sums;     // every sum below was written by the layer in front of it (fusion bit 11)
    for (int i = 0; i < 3; ++i) {     // Decoder.forward: x = de_convs[i](x + en_outs[4 - i]) (models/gtcrn_micro.py:463-469)
        if (!fs) T_RUN(gtt::add_saved(X, t->enc[2 - i].out, t->dec[i].s, n33, s, t->bf));
        if ((rc = gt_fwd(t, t->dec[i], prm, s))) return rc;
        X = t->dec[i].out;
    }
    if (!fs) T_RUN(gtt::add_saved(X, t->en1.a, t->s3, n33, s, t->bf));
    if ((rc = unit_fwd(t, t->de3, prm, s))) return rc;
    if (!fs) T_RUN(gtt::add_saved(t->de3.a, t->en0.a, t->s4, n65, s, t->bf));
    if ((rc = unit_fwd(t, t->de4, prm, s))) return rc;
    T_RUN(gtt::bs_mask_fwd(t->de4.a, d_spec, sb, sf, st, B, T, prm + P(t, "erb.ierb_fc.weight"), d_out, ob, of, ot, s,
                           t->bf));
    t->have_fwd = true;
    t->shift_ready = true;
    return 0;
}

int gtcrn_train_backward(gtcrn_trainer* t, const float* d_params, const float* d_spec, long sb, long sf, long st,
                         const float* d_grad_out, long gb, long gf, long gt, float* d_grads, void* stream) {
    if (!t || !d_params || !d_spec || !d_grad_out || !d_grads)
        return tfail(GTCRN_ERR_ARG, "gtcrn_train_backward: bad argument");
    if (!t->have_fwd) return tfail(GTCRN_ERR_STATE, "gtcrn_train_backward: no forward pass to differentiate");
    // (checked before anything is enqueued: a refusal leaves the gradient blob and the deferral state untouched)
    if (t->gbf && !(t->fusions & 16))
        return tfail(GTCRN_ERR_STATE, "gtcrn_train_backward: bf16 gradient hand-offs (storage 5) need fusion bit 4 (skip "
                                      "gradients accumulated in place: the separate add passes are fp32)");
    T_HIP(hipSetDevice(t->device));
    hipStream_t s = (hipStream_t)stream;
    const int B = t->B, T = t->T;
    const float* prm = d_params;
    float* G = d_grads;
    int rc;
    if ((rc = ensure_wfin_pool(t))) return rc;
    T_HIP(hipMemsetAsync(G, 0, sizeof(float) * GTCRN_NPARAM_FLOATS, s));
    gtt::set_fin_context((t->fusions & 1024) != 0, t->fin_gpart, t->fin_ctr);
    gtt::set_pointwise_form((t->fusions & 8192) != 0);
    gtt::set_column_form((t->fusions & 16384) != 0);
    T_HIP(hipMemsetAsync(t->fin_ctr, 0, sizeof(unsigned) * gtt::FIN_CTR_WORDS, s));
    // the weight-gradient deferral is thread-local state of the launch layer: switched off again on EVERY way out of
    // this function (an early error return used to leave it on, pointing at this trainer's pool)
    struct DeferGuard {
        ~DeferGuard() { gtt::set_wgrad_defer(false, nullptr, 0); }
    } defer_guard;
    gtt::set_wgrad_defer((t->fusions & 32768) != 0, t->wfin_pool, WFIN_POOL_FLOATS);
    t->red_unit = nullptr;
    const long n33 = (long)B * T * 33 * 16, n65 = (long)B * T * 65 * 16;
    T_RUN(gtt::bs_mask_bwd(d_grad_out, gb, gf, gt, d_spec, sb, sf, st, B, T, prm + P(t, "erb.ierb_fc.weight"), t->dm, s));
    // de_convs.4 <- s4 = de3.a + en0.a : gs0 is the gradient of both addends (the sums are recomputed, see plan())
    if (t->share_sums) T_RUN(gtt::add_saved(t->de3.a, t->en0.a, t->s4, n65, s, t->bf));
    if ((rc = unit_bwd(t, t->de4, prm, G, t->dm, t->gs0, 0, nullptr, 0, s, 0, -1))) return rc;      // (dm: fp32)
    if (t->share_sums) T_RUN(gtt::add_saved(t->dec[2].out, t->en1.a, t->s3, n33, s, t->bf));
    if ((rc = unit_bwd(t, t->de3, prm, G, t->gs0, t->gs[1], 0, nullptr, 0, s))) return rc;   // gs[1]: d s3
    // decoder blocks 2,1,0: d s_i is the gradient of the previous block's output and of the skip en_outs[4-i]
    const float* dout = t->gs[1];
    for (int i = 2; i >= 0; --i) {
        float* ds = t->gs[4 - i];            // gs[2], gs[3], gs[4]
        if (t->share_sums)
            T_RUN(gtt::add_saved(i ? t->dec[i - 1].out : t->tcn[7].c3.a, t->enc[2 - i].out, t->dec[i].s, n33, s, t->bf));
        if ((rc = gt_bwd(t, t->dec[i], prm, G, dout, ds, 0, s))) return rc;
        dout = ds;
    }
    // Every encoder tensor en_outs[j] receives two gradients: the decoder's skip (gs[j], gs0 for j = 0 -- complete by
    // now) and the main path's.  The main-path producer ADDS its part into the skip's buffer (accumulating stores of the
    // kernels that form it) instead of a separate pass that reads both and writes the sum: five adds per step gone.
    // GTCN: out = act3(bn3(conv3(..)) + x): dx = dz3 (residual) + conv1 data gradient
    float* qa = t->q1;
    float* qb = t->q2;
    const bool inplace = (t->fusions & 16) != 0;
    for (int i = 7; i >= 0; --i) {
        TcnBlock& k = t->tcn[i];
        // block 0's input is en_outs[4]: both parts go on top of gs[4] (block 7 read it as its gradient input long ago)
        float* dxk = (i == 0 && inplace) ? t->gs[4] : qa;
        const int acc = (i == 0 && inplace) ? 1 : 0;
        if ((rc = unit_bwd(t, k.c3, prm, G, dout, t->dhd, 0, dxk, acc, s))) return rc;    // dhd: d a2, dxk: residual part
        if ((rc = unit_bwd(t, k.c2, prm, G, t->dhd, t->dh, 0, nullptr, 0, s))) return rc;  // dh: d a1
        if ((rc = unit_bwd(t, k.c1, prm, G, t->dh, dxk, 1, nullptr, 0, s))) return rc;     // += conv1 data gradient
        dout = dxk;
        float* tq = qa; qa = qb; qb = tq;
    }
    if (!inplace) {      // (fusion bit 4 off: the separate adds, total gradient of en_outs[k+2] = main path + decoder skip)
        for (int k = 2; k >= 0; --k) {
            T_RUN(gtt::add(dout, t->gs[k + 2], qa, n33, s));
            if ((rc = gt_bwd(t, t->enc[k], prm, G, qa, qb, 0, s))) return rc;
            dout = qb;
            float* tq = qa; qa = qb; qb = tq;
        }
        T_RUN(gtt::add(dout, t->gs[1], qa, n33, s));                    // d en1.a
        if ((rc = unit_bwd(t, t->en1, prm, G, qa, t->d65, 0, nullptr, 0, s))) return rc;
        T_RUN(gtt::add(t->d65, t->gs0, t->d65, n65, s));                // d en0.a
        if ((rc = unit_bwd(t, t->en0, prm, G, t->d65, t->df0, 0, nullptr, 0, s))) return rc;
    } else {
        // encoder blocks 2,1,0: dout = total gradient of en_outs[k+2]; the block's input gradient lands on gs[k+1]
        for (int k = 2; k >= 0; --k) {
            if ((rc = gt_bwd(t, t->enc[k], prm, G, dout, t->gs[k + 1], 1, s))) return rc;
            dout = t->gs[k + 1];
        }
        if ((rc = unit_bwd(t, t->en1, prm, G, dout, t->gs0, 1, nullptr, 0, s))) return rc;     // gs0 += : d en0.a
        if ((rc = unit_bwd(t, t->en0, prm, G, t->gs0, t->df0, 0, nullptr, 0, s, -1, 0))) return rc;      // (df0: fp32)
    }
    {   // SFE weight gradient (no bias); the ERB bank is frozen and the input is data: the chain ends here
        DwGeom g{};
        g.B = B; g.Tin = T; g.Tout = T; g.F = 129; g.C = 3; g.nkt = 1; g.nkf = 3;
        g.f_off[0] = -1; g.f_off[1] = 0; g.f_off[2] = 1; g.w_c = 3; g.w_kt = 3; g.w_kf = 1;
        g.in_bf = t->bf; g.out_bf = 0;
        T_RUN(gtt::dw_wgrad(g, t->eb, t->df0, G + P(t, "sfe.depth_conv.weight"), nullptr, t->fscratch, s));
    }
    T_RUN(gtt::flush_wgrad_finishes(s));      // (fusion bit 15: the recorded weight-gradient finishes, two launches)
    return 0;
}

int gtcrn_train_loss(gtcrn_trainer* t, const float* d_pred, long pb, long pf, long pt, const float* d_true, long tb,
                     long tf, long tt, int B, int T, float* d_loss, float* d_grad, void* stream) {
    // d_grad contiguous (B,257,T,2): strides (257*T*2, T*2, 2)
    return gtcrn_train_loss_strided(t, d_pred, pb, pf, pt, d_true, tb, tf, tt, B, T, d_loss, d_grad, 257L * T * 2, (long)T * 2, 2,
                                    stream);
}

int gtcrn_train_loss_strided(gtcrn_trainer* t, const float* d_pred, long pb, long pf, long pt, const float* d_true, long tb,
                             long tf, long tt, int B, int T, float* d_loss, float* d_grad, long gb, long gf, long gt,
                             void* stream) {
    if (!t || !d_pred || !d_true || !d_loss || B < 1 || B > 1024 || T < 2)
        return tfail(GTCRN_ERR_ARG, "gtcrn_train_loss: bad argument (needs 1 <= B <= 1024 utterances, T >= 2 frames)");
    if (((pb | pf | pt | tb | tf | tt) & 1) || (reinterpret_cast<uintptr_t>(d_pred) & 7) ||
        (reinterpret_cast<uintptr_t>(d_true) & 7))
        return tfail(GTCRN_ERR_ARG, "gtcrn_train_loss: spectrograms must be 8-byte aligned with even strides");
    if (d_grad && (((gb | gf | gt) & 1) || (reinterpret_cast<uintptr_t>(d_grad) & 7)))
        return tfail(GTCRN_ERR_ARG, "gtcrn_train_loss: the gradient must be 8-byte aligned with even strides");
    T_HIP(hipSetDevice(t->device));
    hipStream_t s = (hipStream_t)stream;
    const long Lw = 256L * (T - 1);
    const size_t need = (size_t)2 * B * Lw + 2 * B + 64;
    if (need > t->loss_ws_floats) {
        T_HIP(hipDeviceSynchronize());
        if (t->loss_ws) (void)hipFree(t->loss_ws);
        t->loss_ws = nullptr; t->loss_ws_floats = 0;
        T_HIP(hipMalloc(&t->loss_ws, need * sizeof(float)));
        t->loss_ws_floats = need;
    }
    if (!t->d_win) {
        float w[512];
        gtcrn::make_window(0, w);      // torch.hann_window(512).pow(0.5), loss.py:50
        T_HIP(hipMalloc(&t->d_win, sizeof(w)));
        T_HIP(hipMemcpy(t->d_win, w, sizeof(w), hipMemcpyHostToDevice));
    }
    float* tw = nullptr;
    if (gtcrn_device_twiddles_(&tw)) return GTCRN_ERR_HIP;
    float* yp = t->loss_ws;
    float* yt = yp + (size_t)B * Lw;
    float* coef = yt + (size_t)B * Lw;
    double* spec_partial = t->dscratch;                      // <= MAX_PARTIALS * 2 doubles
    double* dwork = t->dscratch + 2 * gtt::MAX_PARTIALS;     // B * 25 doubles
    int parts = 0;
    T_RUN(gtt::hybrid_loss_spec(d_pred, pb, pf, pt, d_true, tb, tf, tt, B, T, d_grad, gb, gf, gt, spec_partial, &parts, s));
    T_RUN(gtk::launch_istft(d_pred, pb, pf, pt, B, T, nullptr, t->d_win, tw, yp, s));
    T_RUN(gtk::launch_istft(d_true, tb, tf, tt, B, T, nullptr, t->d_win, tw, yt, s));
    T_RUN(gtt::sisnr_terms(yp, yt, B, Lw, spec_partial, parts, (long)B * 257 * T, t->d_win, dwork, coef, d_loss,
                           d_grad != nullptr, s));
    if (d_grad) T_RUN(gtk::launch_istft_adjoint(yp, B, T, t->d_win, tw, d_grad, gb, gf, gt, s));
    return 0;
}

int gtcrn_train_tap(gtcrn_trainer* t, const char* name, float* d_out, long* shape4, void* stream) {
    if (!t || !name) return tfail(GTCRN_ERR_ARG, "gtcrn_train_tap: bad argument");
    if (!t->have_fwd) return tfail(GTCRN_ERR_STATE, "gtcrn_train_tap: no forward pass yet");
    T_HIP(hipSetDevice(t->device));
    auto it = t->taps.find(name);
    if (it == t->taps.end()) return tfail(GTCRN_ERR_ARG, std::string("gtcrn_train_tap: unknown stage ") + name);
    const std::vector<int>& sh = it->second.second;   // T', F, C
    if (shape4) { shape4[0] = t->B; shape4[1] = sh[0]; shape4[2] = sh[1]; shape4[3] = sh[2]; }
    if (d_out) {
        // (fusion bit 11: a block output that was stored only as block output + skip is recovered by subtracting the skip
        // -- equal to the unfused tap up to the rounding of the sum)
        auto mi = t->tap_minus.find(name);
        T_RUN(gtt::saved_to_f32(it->second.first, d_out, (long)t->B * sh[0] * sh[1] * sh[2], (hipStream_t)stream, t->bf,
                                mi == t->tap_minus.end() ? nullptr : mi->second));
    }
    return 0;
}

long gtcrn_clip_adam_workspace_bytes(long n) { return n < 1 ? -1 : 64 + 8 * ((n + 255) / 256); }

int gtcrn_clip_adam_step(int device, float* d_params, float* d_grads, float* d_exp_avg, float* d_exp_avg_sq,
                         const float* d_mask, long n, float max_norm, double lr, double beta1, double beta2, double eps,
                         double weight_decay, long step, float* d_norm_out, void* d_workspace, void* stream) {
    if (!d_params || !d_grads || !d_exp_avg || !d_exp_avg_sq || !d_mask || !d_workspace || n < 1 || n > (1L << 30) ||
        step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(lr >= 0.0) ||
        (reinterpret_cast<uintptr_t>(d_workspace) & 7))
        return tfail(GTCRN_ERR_ARG, "gtcrn_clip_adam_step: bad argument (needs device pointers, an 8-byte aligned "
                                    "workspace, 1 <= n, step >= 1, 0 <= beta < 1, eps >= 0, lr >= 0)");
    struct DeviceScope {            // a C caller's current device is put back on every way out
        int prev = -1;
        DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
        ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    } device_scope;
    T_HIP(hipSetDevice(device));
    // workspace: [0] the last-workgroup ticket (0 between calls), [8..16) norm / coefficient when the caller wants
    // neither, [64..) one double per workgroup of the norm
    char* ws = static_cast<char*>(d_workspace);
    unsigned* counter = reinterpret_cast<unsigned*>(ws);
    float* norm = d_norm_out ? d_norm_out : reinterpret_cast<float*>(ws + 8);
    double* partial = reinterpret_cast<double*>(ws + 64);
    // torch.optim.Adam (_single_tensor_adam, not capturable): the bias corrections are Python floats (double)
    const double bc1 = 1.0 - std::pow(beta1, (double)step);
    const double bc2 = 1.0 - std::pow(beta2, (double)step);
    T_RUN(gtt::clip_adam(d_params, d_grads, d_exp_avg, d_exp_avg_sq, d_mask, (int)n, max_norm, (float)beta1, (float)beta2,
                         (float)(1.0 - beta1), (float)(1.0 - beta2), (float)(lr / bc1), (float)std::sqrt(bc2), (float)eps,
                         (float)weight_decay, norm, partial, counter, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
