// pack.cpp -- host side of the weight contract: the reference state_dict blob
// (canonical order, see include/gtcrn_micro_hip.h) -> BatchNorm folded, slot-space
// packed buffers the gfx950 kernels consume (layout.h).  Pure C++17, no HIP.
//
// Reference semantics restated here (paths relative to the reference repo):
//   * eval-mode BatchNorm2d folding: y = (conv(x) - mean) * gamma / sqrt(var + 1e-5) + beta
//     (ConvBlock.forward models/gtcrn_micro.py:163-164, GTConvBlock :233-243, TCN :296-310)
//   * Conv2d weight (out,in,kt,kf); ConvTranspose2d weight (in,out,kt,kf)
//     (streaming/conversion/convert.py:35-48)
//   * GTConvBlock split/shuffle: x1 = channels 0..7, x2 = 8..15, out[2c] = h[c],
//     out[2c+1] = x2[c] (models/gtcrn_micro.py:222-253)
#include "pack.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "layout.h"

namespace gtcrn {
using namespace gtl;

// ------------------------------------------------------------ parameter table
static void add(std::vector<ParamInfo>& v, const std::string& name, long numel) {
    long off = v.empty() ? 0 : v.back().offset + v.back().numel;
    v.push_back({name, numel, off});
}
static void add_bn(std::vector<ParamInfo>& v, const std::string& p, int C) {
    add(v, p + ".weight", C); add(v, p + ".bias", C);
    add(v, p + ".running_mean", C); add(v, p + ".running_var", C);
}
static void add_convblock(std::vector<ParamInfo>& v, const std::string& p, long wnumel, int Cout, bool act) {
    add(v, p + ".conv.weight", wnumel); add(v, p + ".conv.bias", Cout);
    add_bn(v, p + ".bn", Cout);
    if (act) add(v, p + ".act.weight", 1);
}
static void add_gtconv(std::vector<ParamInfo>& v, const std::string& p, bool dense) {
    add(v, p + ".point_conv1.weight", 128); add(v, p + ".point_conv1.bias", 16);
    add_bn(v, p + ".point_bn1", 16); add(v, p + ".point_act.weight", 1);
    add(v, p + ".depth_conv.weight", dense ? 2304 : 144); add(v, p + ".depth_conv.bias", 16);
    add_bn(v, p + ".depth_bn", 16); add(v, p + ".depth_act.weight", 1);
    add(v, p + ".point_conv2.weight", 128); add(v, p + ".point_conv2.bias", 8);
    add_bn(v, p + ".point_bn2", 8);
    add(v, p + ".tra.depth_conv.weight", 24); add(v, p + ".tra.depth_conv.bias", 8);
    add(v, p + ".tra.point_conv.weight", 64); add(v, p + ".tra.point_conv.bias", 8);
}

const std::vector<ParamInfo>& param_table() {
    static const std::vector<ParamInfo> table = [] {
        std::vector<ParamInfo> v;
        add(v, "erb.erb_fc.weight", 64 * 192);
        add(v, "erb.ierb_fc.weight", 192 * 64);
        add(v, "sfe.depth_conv.weight", 9);
        add_convblock(v, "encoder.en_convs.0", 16 * 3 * 5, 16, true);
        add_convblock(v, "encoder.en_convs.1", 16 * 16 * 5, 16, true);
        for (int k = 2; k < 5; ++k) add_gtconv(v, "encoder.en_convs." + std::to_string(k), false);
        for (int g = 1; g <= 2; ++g)
            for (int k = 0; k < 4; ++k) {
                std::string p = "gtcn" + std::to_string(g) + ".blocks." + std::to_string(k);
                add(v, p + ".conv1.weight", 256); add(v, p + ".conv1.bias", 16);
                add_bn(v, p + ".bn1", 16); add(v, p + ".act1.weight", 1);
                add(v, p + ".conv2.weight", 48); add(v, p + ".conv2.bias", 16);
                add_bn(v, p + ".bn2", 16); add(v, p + ".act2.weight", 1);
                add(v, p + ".conv3.weight", 256); add(v, p + ".conv3.bias", 16);
                add_bn(v, p + ".bn3", 16); add(v, p + ".act3.weight", 1);
            }
        for (int k = 0; k < 3; ++k) add_gtconv(v, "decoder.de_convs." + std::to_string(k), true);
        add_convblock(v, "decoder.de_convs.3", 16 * 16 * 5, 16, true);
        add_convblock(v, "decoder.de_convs.4", 16 * 2 * 5, 2, false);
        return v;
    }();
    return table;
}

// ------------------------------------------------------------------- helpers
namespace {

struct Cursor {
    const float* p;
    const float* take(long n) { const float* r = p; p += n; return r; }
};
struct BN { const float *w, *b, *rm, *rv; };
static BN take_bn(Cursor& c, int C) { BN b; b.w = c.take(C); b.b = c.take(C); b.rm = c.take(C); b.rv = c.take(C); return b; }

// folded per-channel affine: y = conv_nobias(x) * scale + shift
static void fold(const BN& bn, const float* conv_bias, int C, std::vector<double>& scale, std::vector<double>& shift) {
    scale.resize(C); shift.resize(C);
    for (int o = 0; o < C; ++o) {
        double s = (double)bn.w[o] / std::sqrt((double)bn.rv[o] + 1e-5);
        scale[o] = s;
        shift[o] = ((double)(conv_bias ? conv_bias[o] : 0.f) - (double)bn.rm[o]) * s + (double)bn.b[o];
    }
}

struct Perm {          // slot -> logical channel
    int l[16];
    int slot_of(int logical) const { for (int s = 0; s < 16; ++s) if (l[s] == logical) return s; return -1; }
};
static Perm identity() { Perm p; for (int s = 0; s < 16; ++s) p.l[s] = s; return p; }

struct GTRaw {
    const float *pc1_w, *pc1_b; BN bn1; float a1;
    const float *dc_w, *dc_b; BN bn2; float a2;
    const float *pc2_w, *pc2_b; BN bn3;
    const float *tdw, *tdb, *tpw, *tpb;
};
static GTRaw take_gt(Cursor& c, bool dense) {
    GTRaw g;
    g.pc1_w = c.take(128); g.pc1_b = c.take(16); g.bn1 = take_bn(c, 16); g.a1 = *c.take(1);
    g.dc_w = c.take(dense ? 2304 : 144); g.dc_b = c.take(16); g.bn2 = take_bn(c, 16); g.a2 = *c.take(1);
    g.pc2_w = c.take(128); g.pc2_b = c.take(8); g.bn3 = take_bn(c, 8);
    g.tdw = c.take(24); g.tdb = c.take(8); g.tpw = c.take(64); g.tpb = c.take(8);
    return g;
}

// Packs one GTConv block in slot space; in = permutation of the block input, returns output permutation.
// deconv: 1x1 weights are ConvTranspose2d (in,out); dense 3x3 transposed conv packed at GB_DN_A.
static Perm pack_gtconv(const GTRaw& g, bool deconv, const Perm& in, float* F, int* I) {
    std::vector<double> sc, sh;
    // point_conv1 (8 -> 16) + point_bn1; hidden channels keep their own order
    fold(g.bn1, g.pc1_b, 16, sc, sh);
    for (int h = 0; h < 16; ++h) {
        for (int s = 0; s < 16; ++s) {
            int lc = in.l[s];
            double w = 0.0;
            if (lc < 8) w = deconv ? g.pc1_w[lc * 16 + h] : g.pc1_w[h * 8 + lc];
            F[GB_PC1_A + h * 16 + s] = (float)(w * sc[h]);
        }
        F[GB_PC1_B + h] = (float)sh[h];
    }
    // depth conv + depth_bn
    fold(g.bn2, g.dc_b, 16, sc, sh);
    if (!deconv) {
        // depthwise Conv2d (16,1,3,3): out[c,t,f] = sum w[c,kt,kf] * h[c, t-2+kt, f-1+kf]
        for (int kt = 0; kt < 3; ++kt)
            for (int kf = 0; kf < 3; ++kf)
                for (int c = 0; c < 16; ++c)
                    F[GB_DW_W + (kt * 3 + kf) * 16 + c] = (float)((double)g.dc_w[(c * 3 + kt) * 3 + kf] * sc[c]);
    } else {
        for (int i = 0; i < 144; ++i) F[GB_DW_W + i] = 0.f;
        // dense ConvTranspose2d (in,out,3,3): y[o,t,f] = sum_i h[i, t-kt, f+1-kf] * W[i,o,kt,kf]
        for (int kt = 0; kt < 3; ++kt)
            for (int kf = 0; kf < 3; ++kf)
                for (int o = 0; o < 16; ++o)
                    for (int i = 0; i < 16; ++i)
                        F[GB_DN_A + (kt * 3 + kf) * 256 + o * 16 + i] =
                            (float)((double)g.dc_w[((i * 16 + o) * 3 + kt) * 3 + kf] * sc[o]);
    }
    for (int c = 0; c < 16; ++c) F[GB_DW_B + c] = (float)sh[c];
    // point_conv2 (16 -> 8) + point_bn2, written in place over the x1 slots
    fold(g.bn3, g.pc2_b, 8, sc, sh);
    Perm out;
    int c_next = 0;
    for (int s = 0; s < 16; ++s) {
        if (in.l[s] < 8) {  // x1 slot: receives h'[c]
            int c = c_next++;
            for (int h = 0; h < 16; ++h) {
                double w = deconv ? g.pc2_w[h * 8 + c] : g.pc2_w[c * 16 + h];
                F[GB_PC2_A + s * 16 + h] = (float)(w * sc[c]);
            }
            F[GB_PC2_B + s] = (float)sh[c];
            F[GB_KEEP + s] = 0.f;
            I[c] = s;                 // slot_of_c
            out.l[s] = 2 * c;         // shuffle: out[2c] = h'[c]
        } else {            // x2 slot: passes through
            int c = in.l[s] - 8;
            for (int h = 0; h < 16; ++h) F[GB_PC2_A + s * 16 + h] = 0.f;
            F[GB_PC2_B + s] = 0.f;
            F[GB_KEEP + s] = 1.f;
            I[8 + c] = s;             // x2slots (indexed by x2 channel)
            out.l[s] = 2 * c + 1;     // shuffle: out[2c+1] = x2[c]
        }
    }
    std::memcpy(F + GB_TRA_DW, g.tdw, 24 * sizeof(float));
    std::memcpy(F + GB_TRA_DB, g.tdb, 8 * sizeof(float));
    std::memcpy(F + GB_TRA_PW, g.tpw, 64 * sizeof(float));
    std::memcpy(F + GB_TRA_PB, g.tpb, 8 * sizeof(float));
    F[GB_SLOPE + 0] = g.a1; F[GB_SLOPE + 1] = g.a2; F[GB_SLOPE + 2] = 0.f; F[GB_SLOPE + 3] = 0.f;
    return out;
}

// out[s] = index inside a record stored in `consumer` order for the value held in slot s of `producer`
static void store_table(const Perm& producer, const Perm& consumer, int* out) {
    for (int s = 0; s < 16; ++s) out[s] = consumer.slot_of(producer.l[s]);
}

// Every 16x16 A-operand matrix is read by the kernels as one ds_read_b128 per lane: lane (n, g) takes columns
// 4g..4g+3 of row n.  With plain 64-byte rows the eight lanes of one slot group inside a ds_read_b128 lane group
// ({0-3,12-15,20-27}, ...) land on four 16-byte bank slots (2-way conflict).  Rows 4..7 and 12..15 are therefore
// stored with their two 32-byte halves exchanged (quarter q of row n sits at q ^ 2 when bit 2 of n is set), which
// puts the sixteen lanes of every group on sixteen different slots; the kernels read quarter g ^ ((n >> 1) & 2)
// (kernels.hip: arow).  A permutation inside a row: per-row quantisation groups are unaffected.
static void swizzle_matrix(float* M) {
    for (int n = 0; n < 16; ++n)
        if ((n >> 2) & 1)
            for (int c = 0; c < 8; ++c) std::swap(M[n * 16 + c], M[n * 16 + 8 + c]);
}
static void swizzle_matrices(float* F) {
    float* E = F + P_ENC;
    float* D = F + P_DEC;
    swizzle_matrix(E + E_EN0_A);
    for (int k = 0; k < 5; ++k) swizzle_matrix(E + E_EN1_A + k * 256);
    for (int k = 0; k < 3; ++k) {
        swizzle_matrix(E + E_BLK + k * GB_SIZE + GB_PC1_A);
        swizzle_matrix(E + E_BLK + k * GB_SIZE + GB_PC2_A);
    }
    for (int t = 0; t < 8; ++t) {
        swizzle_matrix(F + P_GTCN + t * TCN_SIZE + TCN_A1);
        swizzle_matrix(F + P_GTCN + t * TCN_SIZE + TCN_A3);
    }
    for (int j = 0; j < 3; ++j) {
        float* B = D + D_BLK + j * GBD_SIZE;
        swizzle_matrix(B + GB_PC1_A);
        swizzle_matrix(B + GB_PC2_A);
        for (int t = 0; t < 9; ++t) swizzle_matrix(B + GB_DN_A + t * 256);
    }
    for (int a = 0; a < 3; ++a) swizzle_matrix(D + D_DE3_AE + a * 256);
    for (int a = 0; a < 2; ++a) swizzle_matrix(D + D_DE3_AO + a * 256);
    swizzle_matrix(D + D_DE4_A);
    for (int m = 0; m < 3 * DN16_MATS; ++m) swizzle_matrix(D + D_DN16 + m * 256);
    for (int m = 0; m < DE3_16_MATS; ++m) swizzle_matrix(D + D_DE3_16 + m * 256);
}

// float -> bfloat16 bits, round to nearest even (finite inputs)
static uint16_t bf16_rne(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_to_float(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// w = hi + mid + lo exactly: every residual is formed exactly in fp32 and the last one has at most 8 significant bits
static void split3(float w, uint16_t out[3]) {
    out[0] = bf16_rne(w);
    const float r1 = w - bf16_to_float(out[0]);
    out[1] = bf16_rne(r1);
    const float r2 = r1 - bf16_to_float(out[1]);
    out[2] = bf16_rne(r2);
}

// three bf16 planes of ONE [16 rows][32 k] A-operand matrix whose K halves are the (unswizzled) fp32 slot matrices lo16
// (k < 16) and hi16 (k >= 16; nullptr = zero): out = [3 planes][256 "floats"]
static void pack_planes_pair(const float* lo16, const float* hi16, float* out) {
    uint16_t* H = reinterpret_cast<uint16_t*>(out);
    for (int o = 0; o < 16; ++o)
        for (int k = 0; k < 32; ++k) {
            const float* m = (k >> 4) ? hi16 : lo16;
            uint16_t pl3[3] = {0, 0, 0};
            if (m) split3(m[o * 16 + (k & 15)], pl3);
            for (int p = 0; p < 3; ++p) H[(p * 256) * 2 + o * 32 + k] = pl3[p];
        }
}
// the bf16 planes of one decoder block's dense 3x3 (layout.h, D_DN16) from its (unswizzled) fp32 slot matrices
static void pack_dense_planes(const float* dn32, float* dn16) {
    for (int c = 0; c < DN16_CHUNKS; ++c)
        pack_planes_pair(dn32 + (2 * c) * 256, 2 * c + 1 < 9 ? dn32 + (2 * c + 1) * 256 : nullptr, dn16 + c * 3 * 256);
}

}  // namespace

int pack_params(const float* params, long n, float* F, int* I, std::string& err) {
    if (!params || n != NPARAM) { err = "parameter blob must hold 44938 floats"; return -1; }
    std::memset(F, 0, sizeof(float) * P_FLOATS);
    std::memset(I, 0, sizeof(int) * P_INTS);
    Cursor c{params};
    std::vector<double> sc, sh;

    // ---- ERB (fixed filterbank): contiguous supports -> banded tables ----------------
    const float* erb = c.take(64 * 192);   // erb_fc.weight (64,192)
    const float* ierb = c.take(192 * 64);  // ierb_fc.weight (192,64)
    float* E = F + P_ENC;
    float* D = F + P_DEC;
    for (int j = 0; j < 64; ++j) {
        int lo = -1, hi = -1;
        for (int i = 0; i < 192; ++i) if (erb[j * 192 + i] != 0.f) { if (lo < 0) lo = i; hi = i; }
        if (lo < 0) { lo = 0; hi = -1; }
        if (hi - lo + 1 > ERB_MAXBW) { err = "erb_fc band wider than ERB_MAXBW"; return -1; }
        I[I_ERB_LO + j] = lo; I[I_ERB_N + j] = hi - lo + 1;
        for (int i = lo; i <= hi; ++i) E[E_ERB_W + j * ERB_MAXBW + (i - lo)] = erb[j * 192 + i];
    }
    for (int i = 0; i < 192; ++i) {
        int lo = -1, hi = -1;
        for (int j = 0; j < 64; ++j) if (ierb[i * 64 + j] != 0.f) { if (lo < 0) lo = j; hi = j; }
        if (lo < 0) { lo = 0; hi = -1; }
        if (hi - lo + 1 > ERB_MAXBS) { err = "ierb_fc row wider than ERB_MAXBS"; return -1; }
        I[I_BS_LO + i] = lo; I[I_BS_N + i] = hi - lo + 1;
        for (int j = lo; j <= hi; ++j) D[D_BS_W + i * ERB_MAXBS + (j - lo)] = ierb[i * 64 + j];
    }
    // ---- SFE_Lite ---------------------------------------------------------------------
    std::memcpy(E + E_SFE_W, c.take(9), 9 * sizeof(float));
    // ---- en_convs.0: Conv2d(3,16,(1,5)) + BN + PReLU; im2col column e = c*5 + k -----------
    {
        const float* w = c.take(240); const float* b = c.take(16); BN bn = take_bn(c, 16);
        fold(bn, b, 16, sc, sh);
        for (int o = 0; o < 16; ++o) {
            for (int e = 0; e < 15; ++e) E[E_EN0_A + o * 16 + e] = (float)((double)w[o * 15 + e] * sc[o]);
            E[E_EN0_A + o * 16 + 15] = 0.f;
            E[E_EN0_B + o] = (float)sh[o];
        }
        E[E_EN0_S] = *c.take(1);
    }
    // ---- en_convs.1: Conv2d(16,16,(1,5)) + BN + PReLU; one slot matrix per tap -------------
    {
        const float* w = c.take(1280); const float* b = c.take(16); BN bn = take_bn(c, 16);
        fold(bn, b, 16, sc, sh);
        for (int k = 0; k < 5; ++k)
            for (int o = 0; o < 16; ++o)
                for (int i = 0; i < 16; ++i)
                    E[E_EN1_A + k * 256 + o * 16 + i] = (float)((double)w[(o * 16 + i) * 5 + k] * sc[o]);
        for (int o = 0; o < 16; ++o) E[E_EN1_B + o] = (float)sh[o];
        E[E_EN1_S] = *c.take(1);
    }
    Perm perm[9];
    perm[0] = identity();  // en0
    perm[1] = identity();  // en1
    for (int k = 0; k < 3; ++k) {
        GTRaw g = take_gt(c, false);
        perm[2 + k] = pack_gtconv(g, false, perm[1 + k], E + E_BLK + k * GB_SIZE, I + I_ENC_BLK + k * 16);
    }
    // ---- GTCN x2: residual blocks keep the slot order of en4 ---------------------------------
    const Perm& pg = perm[4];
    for (int g = 0; g < 2; ++g)
        for (int k = 0; k < 4; ++k) {
            float* T = F + P_GTCN + g * GTCN_SIZE + k * TCN_SIZE;
            const float* w1 = c.take(256); const float* b1 = c.take(16); BN bn1 = take_bn(c, 16); float a1 = *c.take(1);
            const float* w2 = c.take(48);  const float* b2 = c.take(16); BN bn2 = take_bn(c, 16); float a2 = *c.take(1);
            const float* w3 = c.take(256); const float* b3 = c.take(16); BN bn3 = take_bn(c, 16); float a3 = *c.take(1);
            fold(bn1, b1, 16, sc, sh);
            for (int h = 0; h < 16; ++h) {
                for (int s = 0; s < 16; ++s) T[TCN_A1 + h * 16 + s] = (float)((double)w1[h * 16 + pg.l[s]] * sc[h]);
                T[TCN_B1 + h] = (float)sh[h];
            }
            fold(bn2, b2, 16, sc, sh);
            for (int kk = 0; kk < 3; ++kk)
                for (int h = 0; h < 16; ++h) T[TCN_DW + kk * 16 + h] = (float)((double)w2[h * 3 + kk] * sc[h]);
            for (int h = 0; h < 16; ++h) T[TCN_B2 + h] = (float)sh[h];
            fold(bn3, b3, 16, sc, sh);
            for (int s = 0; s < 16; ++s) {
                int o = pg.l[s];
                for (int h = 0; h < 16; ++h) T[TCN_A3 + s * 16 + h] = (float)((double)w3[o * 16 + h] * sc[o]);
                T[TCN_B3 + s] = (float)sh[o];
            }
            T[TCN_SLOPE + 0] = a1; T[TCN_SLOPE + 1] = a2; T[TCN_SLOPE + 2] = a3; T[TCN_SLOPE + 3] = 0.f;
        }
    // ---- decoder GTConv blocks (input of block j = x + en_outs[4-j]) -----------------------------
    Perm cur = perm[4];
    for (int j = 0; j < 3; ++j) {
        GTRaw g = take_gt(c, true);
        cur = pack_gtconv(g, true, cur, D + D_BLK + j * GBD_SIZE, I + I_DEC_BLK + j * 16);
        perm[5 + j] = cur;
        pack_dense_planes(D + D_BLK + j * GBD_SIZE + GB_DN_A, D + D_DN16 + j * DN16_SIZE);
    }
    // the skip added to the OUTPUT of decoder block j is en_outs[3-j]: en3 after de0, en2 after de1,
    // en1 after de2.  The encoder stores each of them in that consumer's slot order.
    for (int q = 0; q < 3; ++q)             // q = 0: en1 -> de2 order, 1: en2 -> de1, 2: en3 -> de0
        store_table(perm[1 + q], perm[7 - q], I + I_ENST + q * 16);
    // ---- de_convs.3: ConvTranspose2d(16,16,(1,5),stride 2) + BN + PReLU, gather form -------------
    {
        const float* w = c.take(1280); const float* b = c.take(16); BN bn = take_bn(c, 16);
        fold(bn, b, 16, sc, sh);
        const int ke[3] = {0, 2, 4}, ko[2] = {1, 3};
        for (int o = 0; o < 16; ++o) {
            for (int s = 0; s < 16; ++s) {
                int i = cur.l[s];
                for (int a = 0; a < 3; ++a) D[D_DE3_AE + a * 256 + o * 16 + s] = (float)((double)w[(i * 16 + o) * 5 + ke[a]] * sc[o]);
                for (int a = 0; a < 2; ++a) D[D_DE3_AO + a * 256 + o * 16 + s] = (float)((double)w[(i * 16 + o) * 5 + ko[a]] * sc[o]);
            }
            D[D_DE3_B + o] = (float)sh[o];
        }
        D[D_DE3_S] = *c.take(1);
        perm[8] = identity();
        // the split form (layout.h D_DE3_16): E0 = even outputs from bins (m+1 | m), O0 = odd outputs from the same two
        // bins, E1 = even outputs from bin m-1
        pack_planes_pair(D + D_DE3_AE + 0 * 256, D + D_DE3_AE + 1 * 256, D + D_DE3_16 + 0 * 3 * 256);
        pack_planes_pair(D + D_DE3_AO + 0 * 256, D + D_DE3_AO + 1 * 256, D + D_DE3_16 + 1 * 3 * 256);
        pack_planes_pair(D + D_DE3_AE + 2 * 256, nullptr, D + D_DE3_16 + 2 * 3 * 256);
    }
    // ---- de_convs.4: ConvTranspose2d(16,2,(1,5),stride 2) + BN (+Tanh), scatter form ----------------
    {
        const float* w = c.take(160); const float* b = c.take(2); BN bn = take_bn(c, 2);
        fold(bn, b, 2, sc, sh);
        for (int o = 0; o < 2; ++o) {
            for (int k = 0; k < 5; ++k)
                for (int i = 0; i < 16; ++i)
                    D[D_DE4_A + (o * 5 + k) * 16 + i] = (float)((double)w[(i * 2 + o) * 5 + k] * sc[o]);
            D[D_DE4_B + o] = (float)sh[o];
        }
    }
    if (c.p - params != NPARAM) { err = "internal: parameter walk ended at the wrong offset"; return -1; }
    // ERB.bs gather table (models/gtcrn_micro.py:69-73 as the kernels evaluate it)
    for (int f = 0; f < NBINS; ++f) {
        int i0 = f;
        float w0 = 1.f, w1 = 0.f;
        if (f >= ERB_LOW) {
            const int i = f - ERB_LOW, cnt = I[I_BS_N + i];
            i0 = ERB_LOW + I[I_BS_LO + i];
            w0 = cnt > 0 ? D[D_BS_W + i * ERB_MAXBS] : 0.f;
            w1 = cnt > 1 ? D[D_BS_W + i * ERB_MAXBS + 1] : 0.f;
        }
        float fi;
        std::memcpy(&fi, &i0, 4);
        D[D_BS_TAB + f * 4 + 0] = fi; D[D_BS_TAB + f * 4 + 1] = w0; D[D_BS_TAB + f * 4 + 2] = w1; D[D_BS_TAB + f * 4 + 3] = 0.f;
    }
    Perm stored[9];
    for (int t = 0; t < 9; ++t) stored[t] = perm[t];
    for (int q = 0; q < 3; ++q) stored[1 + q] = perm[7 - q];
    for (int t = 0; t < 9; ++t)
        for (int s = 0; s < 16; ++s) I[I_PERM + t * 16 + s] = stored[t].l[s];
    swizzle_matrices(F);
    return 0;
}

// ------------------------------------------------------------ int8-weight variant (BASELINE configs[4])
// float -> IEEE binary16 (round to nearest even, subnormals kept) -> float: what v_cvt_f16_f32 does on the GPU
float round_to_half(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return x;                       // inf / nan
    if (a >= 0x477FF000u) {                               // >= 65520: rounds to infinity
        a = 0x7F800000u;
    } else if (a < 0x38800000u) {                         // below the smallest normal half (2^-14): subnormal grid 2^-24
        float f;
        std::memcpy(&f, &a, 4);
        f = std::nearbyint(f * 16777216.0f) * (1.0f / 16777216.0f);   // default rounding mode: to nearest even
        std::memcpy(&a, &f, 4);
    } else {
        const uint32_t lsb = (a >> 13) & 1u;
        a = (a + 0xFFFu + lsb) & ~0x1FFFu;                // keep 10 mantissa bits, ties to even
    }
    a |= sign;
    float r;
    std::memcpy(&r, &a, 4);
    return r;
}

namespace {
// One quantisation group per output channel r: elements base[m*ms + r*rs + c*cs], m < nm, c < nc.
// Symmetric per-channel int8 (onnx2tf -qt per-channel, scripts/onnx2tf.sh:57): scale = max|w| / 127,
// q = round(w / scale) in [-127, 127]; the kernels consume fp16(q * scale).
void quant_group(float* base, int rows, int rs, int nm, int ms, int nc, int cs) {
    for (int r = 0; r < rows; ++r) {
        double mx = 0.0;
        for (int m = 0; m < nm; ++m)
            for (int c = 0; c < nc; ++c) mx = std::max(mx, (double)std::fabs(base[m * ms + r * rs + c * cs]));
        if (mx == 0.0) continue;
        const float scale = (float)(mx / 127.0);
        for (int m = 0; m < nm; ++m)
            for (int c = 0; c < nc; ++c) {
                float& w = base[m * ms + r * rs + c * cs];
                double q = std::nearbyint((double)w / (double)scale);
                q = q > 127.0 ? 127.0 : (q < -127.0 ? -127.0 : q);
                w = round_to_half((float)(q * (double)scale));
            }
    }
}
void quant_gtconv(float* B, bool dense) {
    quant_group(B + GB_PC1_A, 16, 16, 1, 0, 16, 1);
    if (dense) quant_group(B + GB_DN_A, 16, 16, 9, 256, 16, 1);
    else quant_group(B + GB_DW_W, 16, 1, 1, 0, 9, 16);           // [tap][channel]: one scale per channel
    quant_group(B + GB_PC2_A, 16, 16, 1, 0, 16, 1);
    quant_group(B + GB_TRA_DW, 8, 3, 1, 0, 3, 1);
    quant_group(B + GB_TRA_PW, 8, 8, 1, 0, 8, 1);
}
}  // namespace

namespace {
// IEEE binary16 bits of a float that IS a binary16 number (round_to_half's output): exact, no rounding involved
uint16_t half_bits_exact(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7FFFFFFFu;
    if (a == 0) return sign;
    if (a >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((a & 0x007FFFFFu) ? 0x200u : 0u));
    const int e = (int)(a >> 23) - 127;                       // unbiased exponent
    const uint32_t man = (a & 0x007FFFFFu) | 0x00800000u;     // 24-bit significand
    if (e >= -14) return (uint16_t)(sign | (uint32_t)((e + 15) << 10) | ((man >> 13) & 0x3FFu));
    return (uint16_t)(sign | (man >> (13 + (-14 - e))));      // subnormal half: shift the significand further down
}
// The fp16 form of one decoder block's dense 3x3 for v_mfma_f32_16x16x32_f16 (kernels.hip, the Q variant's dense
// phase): 5 K-chunks (taps 2c | 2c+1) of a [16 rows][32 k] fp16 matrix, 256 "floats" each, written over the head of
// the block's D_DN16 slot (the bf16 planes there belong to the fp32 model and are not read in this variant).
// dn32: the block's nine quantised fp32 slot matrices, ALREADY row-swizzled (swizzle_matrix).
void pack_dense_half(const float* dn32, float* dst) {
    for (int c = 0; c < DN16_CHUNKS; ++c) {
        uint16_t* H = reinterpret_cast<uint16_t*>(dst + c * 256);
        for (int o = 0; o < 16; ++o)
            for (int k = 0; k < 32; ++k) {
                const int tap = 2 * c + (k >> 4), i = k & 15;
                float w = 0.f;
                if (tap < 9) w = dn32[tap * 256 + o * 16 + (((o >> 2) & 1) ? (i ^ 8) : i)];     // undo the swizzle
                H[o * 32 + k] = half_bits_exact(w);
            }
        // same row swizzle as every A-operand matrix (on the 16 x 16-float view)
        float* M = dst + c * 256;
        for (int n = 0; n < 16; ++n)
            if ((n >> 2) & 1)
                for (int q = 0; q < 8; ++q) std::swap(M[n * 16 + q], M[n * 16 + 8 + q]);
    }
}
}  // namespace

// In place on the packed float buffer (after BatchNorm folding, which is what the exported graph holds): every
// conv / linear weight becomes fp16(int8 * per-output-channel scale); biases, PReLU slopes and the KEEP masks stay.
void quantize_packed(float* F) {
    float* E = F + P_ENC;
    float* D = F + P_DEC;
    quant_group(E + E_ERB_W, 64, ERB_MAXBW, 1, 0, ERB_MAXBW, 1);
    quant_group(E + E_SFE_W, 3, 3, 1, 0, 3, 1);
    quant_group(E + E_EN0_A, 16, 16, 1, 0, 16, 1);
    quant_group(E + E_EN1_A, 16, 16, 5, 256, 16, 1);
    for (int k = 0; k < 3; ++k) quant_gtconv(E + E_BLK + k * GB_SIZE, false);
    for (int t = 0; t < 8; ++t) {
        float* T = F + P_GTCN + t * TCN_SIZE;
        quant_group(T + TCN_A1, 16, 16, 1, 0, 16, 1);
        quant_group(T + TCN_DW, 16, 1, 1, 0, 3, 16);
        quant_group(T + TCN_A3, 16, 16, 1, 0, 16, 1);
    }
    for (int j = 0; j < 3; ++j) {
        quant_gtconv(D + D_BLK + j * GBD_SIZE, true);
        pack_dense_half(D + D_BLK + j * GBD_SIZE + GB_DN_A, D + D_DN16 + j * DN16_SIZE);
    }
    quant_group(D + D_DE3_AE, 16, 16, 5, 256, 16, 1);            // AE (3 matrices) + AO (2) share the output channels
    quant_group(D + D_DE4_A, 2, 80, 5, 16, 16, 1);               // rows o*5+k: output channel o owns five rows
    quant_group(D + D_BS_W, 192, ERB_MAXBS, 1, 0, ERB_MAXBS, 1);
    // the kernel-ready ERB.bs table repeats those weights (slots 1, 2 of the rows of the high bins)
    for (int f = ERB_LOW; f < NBINS; ++f) {
        const int i = f - ERB_LOW;
        if (D[D_BS_TAB + f * 4 + 1] != 0.f) D[D_BS_TAB + f * 4 + 1] = D[D_BS_W + i * ERB_MAXBS];
        if (D[D_BS_TAB + f * 4 + 2] != 0.f) D[D_BS_TAB + f * 4 + 2] = D[D_BS_W + i * ERB_MAXBS + 1];
    }
}

void make_window(int kind, float* w) {
    // torch.hann_window(512) (periodic, fp32: cos(n * float(2pi/512)) * -0.5 + 0.5); kind 0 takes sqrt
    for (int n = 0; n < 512; ++n) {
        float ang = (float)n * (float)(2.0 * M_PI / 512.0);
        float hann = std::cos(ang) * -0.5f + 0.5f;
        w[n] = kind == 0 ? std::sqrt(hann) : hann;
    }
}

}  // namespace gtcrn
