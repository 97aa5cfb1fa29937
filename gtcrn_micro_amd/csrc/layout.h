// layout.h -- packed ("slot-space") parameter layout shared by the host packer
// (pack.cpp) and the gfx950 kernels (kernels.hip).  Plain constants only.
//
// Slot space: every 16-channel activation is held as 16 "slots"; lane
// (n = lane&15, g = lane>>4) of a wave owns slots 4g..4g+3 of position n of a
// 16-position tile, which is exactly the C/D fragment of
// v_mfma_f32_16x16x4_f32 with the weights as the A operand.  A 16x16 slot
// matrix M is stored row-major (out-slot major): lane (n,g) reads its A
// fragment as the float4 at M[n*16 + 4g].  Which logical channel sits in which
// slot (the permutation the GTConv "shuffle" induces, models/gtcrn_micro.py:
// 222-227) is resolved here on the host, so the kernels never move channels.
#pragma once

namespace gtl {

constexpr int NBINS = 257;  // STFT bins
constexpr int F0 = 129;     // ERB features (65 linear + 64 bands)
constexpr int F1 = 65;      // after en_conv0
constexpr int F2 = 33;      // after en_conv1
constexpr int NPARAM = 44938;
constexpr int NTENSORS = 342;

// ---- ERB tables (models/gtcrn_micro.py:35-73) ------------------------------
constexpr int ERB_LOW = 65;     // pass-through bins
constexpr int ERB_BANDS = 64;
constexpr int ERB_HIGH = 192;   // bins 65..256
constexpr int ERB_MAXBW = 12;   // max taps of one band (11 for the shipped bank)
constexpr int ERB_MAXBS = 2;    // max bands touching one bin

// ---- TCN block (models/gtcrn_micro.py:256-310), floats ----------------------
constexpr int TCN_A1 = 0;       // 16x16 conv1 (rows: hidden, cols: input slots)
constexpr int TCN_B1 = 256;
constexpr int TCN_DW = 272;     // [3 taps][16 hidden]; tap k multiplies y1[t-2d+k*d]
constexpr int TCN_B2 = 320;
constexpr int TCN_A3 = 336;     // 16x16 conv3 (rows: output slots, cols: hidden)
constexpr int TCN_B3 = 592;
constexpr int TCN_SLOPE = 608;  // a1, a2, a3, pad
constexpr int TCN_SIZE = 612;
constexpr int GTCN_SIZE = 4 * TCN_SIZE;

// ---- GTConv block, encoder (depthwise 3x3) ---------------------------------
constexpr int GB_PC1_A = 0;
constexpr int GB_PC1_B = 256;
constexpr int GB_DW_W = 272;    // encoder: [9 taps kt*3+kf][16]; decoder: unused
constexpr int GB_DW_B = 416;
constexpr int GB_PC2_A = 432;
constexpr int GB_PC2_B = 688;
constexpr int GB_KEEP = 704;    // 1.0 for pass-through (x2) slots, 0.0 for slots receiving h'
constexpr int GB_TRA_DW = 720;  // [8][3]
constexpr int GB_TRA_DB = 744;
constexpr int GB_TRA_PW = 752;  // [8][8]
constexpr int GB_TRA_PB = 816;
constexpr int GB_SLOPE = 824;   // a1, a2, pad, pad
constexpr int GB_SIZE = 828;
// decoder blocks append the dense transposed 3x3: 9 slot matrices, tap index kt*3+kf
constexpr int GB_DN_A = GB_SIZE;              // 9*256
constexpr int GBD_SIZE = GB_SIZE + 9 * 256;   // 3132

// ---- encoder segment (floats) ----------------------------------------------
constexpr int E_ERB_W = 0;                         // [64][ERB_MAXBW]
constexpr int E_SFE_W = E_ERB_W + 64 * ERB_MAXBW;  // [3][3] (+3 pad)
constexpr int E_EN0_A = E_SFE_W + 12;              // 16x16: col e = c*5+k (15 used)
constexpr int E_EN0_B = E_EN0_A + 256;
constexpr int E_EN0_S = E_EN0_B + 16;              // slope (+3 pad)
constexpr int E_EN1_A = E_EN0_S + 4;               // [5 taps][16x16]
constexpr int E_EN1_B = E_EN1_A + 5 * 256;
constexpr int E_EN1_S = E_EN1_B + 16;
constexpr int E_BLK = E_EN1_S + 4;                 // 3 x GB_SIZE
constexpr int ENC_SIZE = E_BLK + 3 * GB_SIZE;

// ---- decoder segment (floats) ----------------------------------------------
constexpr int D_BLK = 0;                           // 3 x GBD_SIZE
constexpr int D_DE3_AE = D_BLK + 3 * GBD_SIZE;     // even outputs: taps k=0,2,4 (input f=m+1,m,m-1)
constexpr int D_DE3_AO = D_DE3_AE + 3 * 256;       // odd outputs: taps k=1,3 (input f=m+1,m)
constexpr int D_DE3_B = D_DE3_AO + 2 * 256;
constexpr int D_DE3_S = D_DE3_B + 16;
constexpr int D_DE4_A = D_DE3_S + 4;               // 16x16: row o*5+k (10 used), cols = de3 channels
constexpr int D_DE4_B = D_DE4_A + 256;             // 2 (+2 pad)
constexpr int D_BS_W = D_DE4_B + 4;                // [192][ERB_MAXBS]
// ERB.bs as a uniform 2-tap gather per output bin, ready for the kernel: [257][4] = {first input index (int bits),
// w0, w1, 0}: the 65 low bins pass through (index = bin, weights 1, 0), the others combine at most two bands
constexpr int D_BS_TAB = D_BS_W + 192 * ERB_MAXBS;
// The dense transposed 3x3 of the three decoder blocks for the 16-bit matrix pipe (kernels.hip, "split" form): every
// folded fp32 weight is split EXACTLY into three bf16 planes (w = hi + mid + lo, 3 x 8 significant bits = fp32's 24)
// and laid out as the A operand of v_mfma_f32_16x16x32_bf16: per block 5 K-chunks (chunk c = taps 2c and 2c+1; the
// second half of chunk 4 is zero) x 3 planes of a [16 output rows][32 k] bf16 matrix, k = (tap - 2c) * 16 + hidden
// channel; a matrix is 256 "floats" (bf16 pairs) with the same row swizzle as the fp32 slot matrices (pack.cpp).
// The fp32 matrices at GB_DN_A stay in the buffer: the int8/fp16 variant and the CPU emulation of the packed
// dataflow read them, and the planes are checked against them (tests/test_host_logic.py).
constexpr int DN16_CHUNKS = 5;
constexpr int DN16_MATS = DN16_CHUNKS * 3;
constexpr int DN16_SIZE = DN16_MATS * 256;
constexpr int D_DN16 = D_BS_TAB + NBINS * 4;       // 3 x DN16_SIZE
// de_convs.3 (gather form) in the same split form: K-chunk 0 = input bins (m+1 | m), K-chunk 1 = (m-1 | nothing);
// matrices E0 (even outputs, taps k = 0 | 2), O0 (odd outputs, k = 1 | 3), E1 (even, k = 4 | zero), 3 planes each
constexpr int DE3_16_MATS = 9;
constexpr int D_DE3_16 = D_DN16 + 3 * DN16_SIZE;   // [E0, O0, E1][3 planes][256]
constexpr int DEC_SIZE = D_DE3_16 + DE3_16_MATS * 256;

// ---- whole float buffer ------------------------------------------------------
constexpr int P_ENC = 0;
constexpr int P_GTCN = P_ENC + ENC_SIZE;           // 2 x GTCN_SIZE
constexpr int P_DEC = P_GTCN + 2 * GTCN_SIZE;
constexpr int P_FLOATS = P_DEC + DEC_SIZE;

// ---- int buffer --------------------------------------------------------------
constexpr int I_ERB_LO = 0;                        // [64] first high-bin index (0..191) of band j
constexpr int I_ERB_N = 64;                        // [64] taps of band j
constexpr int I_BS_LO = 128;                       // [192] first band of bin i
constexpr int I_BS_N = 320;                        // [192] bands of bin i
constexpr int I_ENC_BLK = 512;                     // 3 x 16: slot_of_c[8], x2slots[8]
constexpr int I_DEC_BLK = I_ENC_BLK + 48;          // 3 x 16
// encoder store tables (16 each): the skips en1, en2, en3 are stored in the slot order of the
// decoder stage that adds them (so the decoder reads them with one 16-byte load per lane):
// I_ENST[q][s] = index inside the stored 16-float record for the value in encoder slot s.
constexpr int I_ENST = I_DEC_BLK + 48;             // [4][16]: en1, en2, en3, spare
// slot -> logical channel of each STORED activation, for the debug taps:
// 0:en0 1:en1 2:en2 3:en3 4:en4(=gtcn) 5:de0 6:de1 7:de2 8:de3
constexpr int I_PERM = I_ENST + 64;                // [9][16]
constexpr int P_INTS = I_PERM + 9 * 16;

}  // namespace gtl
