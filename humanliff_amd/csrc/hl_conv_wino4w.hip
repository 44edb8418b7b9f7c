// k_conv_wino4w - Winograd F(4x4,3x3) convolution, fp32 on v_mfma_f32_32x32x2_f32, 64 output channels per workgroup, ONE wave per SIMD.
//
// Replaces nn.Conv2d 3x3 / stride 1 (ResBlock in_layers / out_layers, the nearest-x2 Upsample convolution; improved_diffusion/unet.py:52-80,
// 149-166) on the 256- and 128-pixel levels, like k_conv_wino4 (hl_unet_kernels.hip), whose arithmetic, interpolation points (0, +-3/4, +-3/2,
// inf), packed weights and patch layouts it shares.  What differs is the occupancy model:
//
//   k_conv_wino4  : 32 output channels per workgroup, 9 accumulator tiles per wave (144 registers), two workgroups per CU = two waves per
//                   SIMD; a wave alternates a transform phase (VALU + LDS) with an MFMA phase and relies on the partner wave to keep the
//                   matrix pipe busy meanwhile.  Measured: pipe 0.66 busy, 4.6 VALU instructions per MFMA, every 34x18 patch fetched and
//                   transformed by six workgroups (192 output channels).
//   k_conv_wino4w : 64 output channels per workgroup: each of the 4 waves (one per SIMD, the 512-entry register budget) owns its 3x3
//                   frequency block for BOTH 32-channel halves = 18 accumulator tiles in the accumulator registers.  A patch is fetched
//                   and transformed once per 72 MFMAs instead of once per 36, and the transform of k-tile t+1 is software-pipelined into
//                   the 64-cycle shadows of the MFMAs of k-tile t: one transform micro-op (one LDS read, or one 4-wide fma) behind each
//                   MFMA, written out as a fixed 72-slot schedule.  The weight slices go from L2 straight into registers (a ring of six
//                   16-byte fragments, fetched three frequency pairs = 24 MFMAs ahead) - no LDS round trip, no M0 juggling.
//
// Per k-tile (8 input channels) and wave: 72 MFMAs, 25 ds_read_b128, 48 4-wide fmas, 18 buffer_load_dwordx4 (weights), 5-6
// buffer_load_dwordx4 + ds_write_b128 (its share of the next-but-one patch), one barrier.  LDS: two patch stages (2 x 20.25 KB) during the K walk, then the 36 frequencies of 8 tiles x 64
// channels (72 KB) meet there in four rounds for the output transform.
#include <type_traits>

#include "hl_unet_kernels.h"
#include "hl_stats.h"

namespace hl {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float W4_A = 0.75f, W4_B = 1.5f;
constexpr float W4_C0 = W4_A * W4_A * W4_B * W4_B, W4_C2 = -(W4_A * W4_A + W4_B * W4_B);

#ifndef HL_W4W_SCALAR_FMA
#define HL_W4W_SCALAR_FMA 0
#endif
#ifndef HL_W4W_ABL   // timing ablations (wrong results; 256 no epilogue global traffic, 512 no epilogue LDS exchange): 1 no per-tile barrier, 2 no transform, 4 no weight loads, 8 no patch DMA, 16 no MFMA
#define HL_W4W_ABL 0
#endif
// d = c * x + y on four channels.  Packed (v_pk_fma_f32 x2) or four plain v_fma_f32 (HL_W4W_SCALAR_FMA: the instruction selector would
// re-pack scalar fmas, so those are written as asm)
__device__ __forceinline__ f32x4 fma4(float c, f32x4 x, f32x4 y) {
#if HL_W4W_SCALAR_FMA
    f32x4 d;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float r;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x[i]), "s"(c), "v"(y[i]));
        d[i] = r;
    }
    return d;
#else
    return __builtin_elementwise_fma((f32x4)(c), x, y);
#endif
}
__device__ __forceinline__ f32x2 fma2(float c, f32x2 x, f32x2 y) { return __builtin_elementwise_fma((f32x2)(c), x, y); }
// one side of the output transform: four outputs from the six frequencies (0, +a, -a, +b, -b, inf), two channels at once
__device__ __forceinline__ void w4_out(f32x2 m0, f32x2 m1, f32x2 m2, f32x2 m3, f32x2 m4, f32x2 m5, f32x2 (&y)[4]) {
    const f32x2 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y[0] = (m0 + s1) + s2;
    y[1] = fma2(W4_B, d2, W4_A * d1);
    y[2] = fma2(W4_B * W4_B, s2, (W4_A * W4_A) * s1);
    y[3] = fma2(W4_B * W4_B * W4_B, d2, fma2(W4_A * W4_A * W4_A, d1, m5));
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// The 18 accumulator tiles of a wave are 288 registers; the accumulator file holds 256.  Tiles 0..15 ARE the accumulator file, named
// literally (a[16T : 16T+15], declared as clobbers so that the compiler keeps out of it); tiles 16 and 17 are ordinary VGPR values.  With
// compiler-managed accumulators ("+a" operands or the builtin) the register allocator, left with no spare accumulator register, splits a
// tile's live range and copies / spills it next to MFMAs it does not know to be MFMAs (wrong sums), or rotates tiles through the file.
// The instruction is opaque to the compiler, so its hazards are the schedule's business: operands are written many slots before use, an
// accumulator is touched again two MFMAs later at the earliest, and the epilogue waits (s_nop) before reading the tiles.
template <int T>
__device__ __forceinline__ void mfma_tile(f32x16 &c, float a, float b) {
    if constexpr (T == 0) asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    else if constexpr (T == 1) asm volatile("v_mfma_f32_32x32x2_f32 a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b) : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
    else if constexpr (T == 2) asm volatile("v_mfma_f32_32x32x2_f32 a[32:47], %0, %1, a[32:47]" ::"v"(a), "v"(b) : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    else if constexpr (T == 3) asm volatile("v_mfma_f32_32x32x2_f32 a[48:63], %0, %1, a[48:63]" ::"v"(a), "v"(b) : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
    else if constexpr (T == 4) asm volatile("v_mfma_f32_32x32x2_f32 a[64:79], %0, %1, a[64:79]" ::"v"(a), "v"(b) : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
    else if constexpr (T == 5) asm volatile("v_mfma_f32_32x32x2_f32 a[80:95], %0, %1, a[80:95]" ::"v"(a), "v"(b) : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
    else if constexpr (T == 6) asm volatile("v_mfma_f32_32x32x2_f32 a[96:111], %0, %1, a[96:111]" ::"v"(a), "v"(b) : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
    else if constexpr (T == 7) asm volatile("v_mfma_f32_32x32x2_f32 a[112:127], %0, %1, a[112:127]" ::"v"(a), "v"(b) : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
    else if constexpr (T == 8) asm volatile("v_mfma_f32_32x32x2_f32 a[128:143], %0, %1, a[128:143]" ::"v"(a), "v"(b) : "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");
    else if constexpr (T == 9) asm volatile("v_mfma_f32_32x32x2_f32 a[144:159], %0, %1, a[144:159]" ::"v"(a), "v"(b) : "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159");
    else if constexpr (T == 10) asm volatile("v_mfma_f32_32x32x2_f32 a[160:175], %0, %1, a[160:175]" ::"v"(a), "v"(b) : "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175");
    else if constexpr (T == 11) asm volatile("v_mfma_f32_32x32x2_f32 a[176:191], %0, %1, a[176:191]" ::"v"(a), "v"(b) : "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191");
    else if constexpr (T == 12) asm volatile("v_mfma_f32_32x32x2_f32 a[192:207], %0, %1, a[192:207]" ::"v"(a), "v"(b) : "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207");
    else if constexpr (T == 13) asm volatile("v_mfma_f32_32x32x2_f32 a[208:223], %0, %1, a[208:223]" ::"v"(a), "v"(b) : "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223");
    else if constexpr (T == 14) asm volatile("v_mfma_f32_32x32x2_f32 a[224:239], %0, %1, a[224:239]" ::"v"(a), "v"(b) : "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239");
    else if constexpr (T == 15) asm volatile("v_mfma_f32_32x32x2_f32 a[240:255], %0, %1, a[240:255]" ::"v"(a), "v"(b) : "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    else asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void acc_zero() {   // 256 x v_accvgpr_write_b32, one statement per accumulator tile with its sixteen registers declared
    asm volatile(".set hl_w4w_i, 0\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    asm volatile(".set hl_w4w_i, 16\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
    asm volatile(".set hl_w4w_i, 32\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    asm volatile(".set hl_w4w_i, 48\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
    asm volatile(".set hl_w4w_i, 64\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
    asm volatile(".set hl_w4w_i, 80\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
    asm volatile(".set hl_w4w_i, 96\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
    asm volatile(".set hl_w4w_i, 112\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
    asm volatile(".set hl_w4w_i, 128\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");
    asm volatile(".set hl_w4w_i, 144\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159");
    asm volatile(".set hl_w4w_i, 160\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175");
    asm volatile(".set hl_w4w_i, 176\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191");
    asm volatile(".set hl_w4w_i, 192\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207");
    asm volatile(".set hl_w4w_i, 208\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223");
    asm volatile(".set hl_w4w_i, 224\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239");
    asm volatile(".set hl_w4w_i, 240\n\t.rept 16\n\tv_accvgpr_write_b32 a[hl_w4w_i], 0\n\t.set hl_w4w_i, hl_w4w_i+1\n\t.endr" ::: "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
}
template <int T, int R>
__device__ __forceinline__ float acc_get(const f32x16 (&cv)[2]) {
    if constexpr (T < 16) {
        float v;
        asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(T * 16 + R));
        return v;
    } else return cv[T - 16][R];
}

// ---- the schedule of one k-tile (72 MFMA gaps) -------------------------------------------------------------------------------------
// Measured on MI355X with one wave per SIMD (scripts/microbench/mfma_fill.hip): behind a v_mfma_f32_32x32x2_f32 NOTHING of the vector ALU
// hides - the fp32 MFMA runs on the SIMD's fp32 lanes, a v_pk_fma_f32 beside it costs its own 4 cycles plus ~8 cycles for every gap that
// holds any VALU at all (1 / 2 / 4 / 8 v_pk_fma per gap: 76.5 / 80.5 / 88.5 / 104.5 cycles per MFMA against 64.0) - while ds_read_b128 and
// SALU are free, a buffer_load into registers costs ~9 cycles per gap that holds one or two, and an LDS-DMA piece ~100.  Hence:
//   * the transform arithmetic of k-tile t+1 comes in FIVE bursts (one per window column step: vertical transform + horizontal
//     accumulation, 12..30 v_pk_fma each) instead of one or two instructions per gap;
//   * its 25 window reads are spread one per gap (free) into the gaps before the burst that consumes them;
//   * the weight fragments are fetched in three bursts of six loads (ring of six frequency pairs = 12 fragments, 24+ gaps ahead);
//   * the patch of k-tile t+2 is fetched by ONE burst of 5-6 buffer_load_dwordx4 into registers and stored to LDS with ds_write_b128 near
//     the end of the k-tile (same LDS image as the LDS-DMA of k_conv_wino4 produced; out-of-range lanes load zeros).
// Column steps visit the window columns in the order that turns every horizontal sum into one fma chain with three partials per frequency
// row live: frequency columns (0,+a,-a) want window columns 4,2,0,3,1; (+b,-b,inf) want 3,1,4,2,0 (offset by FC).
constexpr int n_acc(int FC, int k) {   // fmas of the horizontal accumulation at column step k
    return FC == 0 ? (k == 1 ? 6 : (k == 2 ? 3 : (k == 4 ? 3 : 0))) : (k == 1 ? 3 : (k == 3 ? 3 : (k == 4 ? 6 : 0)));
}
constexpr int GAP_BURST0 = 8, GAP_BURST_STEP = 8;      // burst k behind MFMA 8 (k + 1)
constexpr int GAP_PATCH_LOAD = 1, GAP_PATCH_STORE = 58; // patch t+2: loads behind MFMA 1, ds_write j behind MFMA 58 + j
constexpr int read_gap(int k, int r) { return (k == 0 ? 0 : GAP_BURST0 + (k - 1) * GAP_BURST_STEP + 1) + r; }   // after the burst that used the column buffer

struct Tr {   // transform state of one wave: the column buffer, the three vertical outputs, the partial sums (they become V)
    f32x4 x[5], o[3], e, t, u, P[3][3];
};

template <int FR, int I>
__device__ __forceinline__ void op_fwd(const f32x4 (&x)[5], Tr &s) {
    if constexpr (FR == 0) {   // rows (0, +a, -a) on d0..d4
        if constexpr (I == 0) s.u = fma4(W4_C2, x[2], x[4]);
        else if constexpr (I == 1) s.o[0] = fma4(W4_C0, x[0], s.u);
        else if constexpr (I == 2) s.e = fma4(-W4_B * W4_B, x[2], x[4]);
        else if constexpr (I == 3) s.t = fma4(-W4_B * W4_B, x[1], x[3]);
        else if constexpr (I == 4) s.o[1] = fma4(W4_A, s.t, s.e);
        else s.o[2] = fma4(-W4_A, s.t, s.e);
    } else {                   // rows (+b, -b, inf) on d1..d5
        if constexpr (I == 0) s.e = fma4(-W4_A * W4_A, x[1], x[3]);
        else if constexpr (I == 1) s.t = fma4(-W4_A * W4_A, x[0], x[2]);
        else if constexpr (I == 2) s.o[0] = fma4(W4_B, s.t, s.e);
        else if constexpr (I == 3) s.o[1] = fma4(-W4_B, s.t, s.e);
        else if constexpr (I == 4) s.u = fma4(W4_C2, x[2], x[4]);
        else s.o[2] = fma4(W4_C0, x[0], s.u);
    }
}
// plain assignments "P = y" of a column step (register renames, no instructions): done with the last F op of the step
template <int FC, int K>
__device__ __forceinline__ void op_assign(Tr &s) {
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) {
        if constexpr (FC == 0) {
            if constexpr (K == 0) s.P[ii][0] = s.o[ii];
            else if constexpr (K == 3) s.P[ii][2] = s.o[ii];
        } else {
            if constexpr (K == 0) s.P[ii][0] = s.o[ii];
            else if constexpr (K == 2) s.P[ii][2] = s.o[ii];
        }
    }
}
template <int FC, int K, int I>
__device__ __forceinline__ void op_acc(Tr &s) {
    if constexpr (FC == 0) {
        if constexpr (K == 1) {
            constexpr int ii = I >> 1;
            if constexpr ((I & 1) == 0) s.P[ii][1] = fma4(-W4_B * W4_B, s.o[ii], s.P[ii][0]);
            else s.P[ii][0] = fma4(W4_C2, s.o[ii], s.P[ii][0]);
        } else if constexpr (K == 2) s.P[I][0] = fma4(W4_C0, s.o[I], s.P[I][0]);
        else s.P[I][2] = fma4(-W4_B * W4_B, s.o[I], s.P[I][2]);                                    // K == 4
    } else {
        if constexpr (K == 1) s.P[I][0] = fma4(-W4_A * W4_A, s.o[I], s.P[I][0]);
        else if constexpr (K == 3) { s.P[I][2] = fma4(W4_C2, s.o[I], s.P[I][2]); s.P[I][1] = s.o[I]; }
        else {                                                                                     // K == 4
            constexpr int ii = I >> 1;
            if constexpr ((I & 1) == 0) s.P[ii][1] = fma4(-W4_A * W4_A, s.o[ii], s.P[ii][1]);
            else s.P[ii][2] = fma4(W4_C0, s.o[ii], s.P[ii][2]);
        }
    }
}
// final combination of frequency row ii: V[3 ii + (0,1,2)]
template <int FC, int I>
__device__ __forceinline__ void op_fin(Tr &s, f32x4 (&V)[9]) {
    constexpr int ii = I >> 1;
    if constexpr (FC == 0) {
        if constexpr ((I & 1) == 0) { V[ii * 3 + 0] = s.P[ii][0]; V[ii * 3 + 1] = fma4(W4_A, s.P[ii][2], s.P[ii][1]); }
        else V[ii * 3 + 2] = fma4(-W4_A, s.P[ii][2], s.P[ii][1]);
    } else {
        if constexpr ((I & 1) == 0) { V[ii * 3 + 2] = s.P[ii][2]; V[ii * 3 + 0] = fma4(W4_B, s.P[ii][1], s.P[ii][0]); }
        else V[ii * 3 + 1] = fma4(-W4_B, s.P[ii][1], s.P[ii][0]);
    }
}

template <bool UPS, bool BLK>
__global__ __launch_bounds__(256, 1) void k_conv_wino4w(const ConvK p) {
#if __HIP_DEVICE_COMPILE__   // device pass only (the host pass of this clang mis-parses large kernel bodies, see k_conv_bf3)
    constexpr int PRW = 36, PROWS = 18, P_REAL = 2 * PROWS * PRW, NP = (P_REAL + 63) / 64, P_F = P_REAL * 4;   // patch: pixels / row, rows, chunks, DMAs, floats
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;           // contiguous runs of (tile block, channel block) per XCD
    const int tb = wi / p.n_nblocks, nb = wi - tb * p.n_nblocks;
    const int n0 = nb * 64;
    const int Hv = UPS ? 2 * p.Hin : p.Hin, Wv = UPS ? 2 * p.Win : p.Win;
    const int bw = Wv >> 5, bh = Hv >> 4;
    const int img = tb / (bw * bh), brem = tb - img * (bw * bh);
    const int y0 = (brem / bw) * 16, x0 = (brem - (brem / bw) * bw) * 32;
    const int nkt = p.Cin >> 3;
    static_assert(!(UPS && BLK), "the upsampling convolution reads a raw NHWC tensor");
    const unsigned pitch4 = BLK ? 32u : (unsigned)p.in_pitch * 4u;
    const int kstep = BLK ? (int)(p.M * 32) : 32;
    const int kt0 = blockIdx.z * p.kt_per;                            // split-K over input channels: this slab's k-tiles
    const int ntiles = min(nkt, kt0 + p.kt_per) - kt0;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.in, (short)0, (int)((long)p.N * p.Hin * p.Win * p.in_pitch * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.w_wino, (short)0, (int)((long)(p.Cout >> 5) * nkt * 36 * 1024), 0x00020000);

#ifndef HL_W4W_STAGGER
#define HL_W4W_STAGGER 0
#endif
    // HL_W4W_STAGGER = n (experiment): the first workgroup of every CU starts (its index / 8 mod 4) * n * 0.9 us late, so that the CUs of an XCD
    // are not all in their epilogues (256 KB of HBM traffic per workgroup) at the same moment.
    if (HL_W4W_STAGGER > 0 && blockIdx.x < 256 && gridDim.x >= 512) {
        const int ph = (blockIdx.x >> 3) & 3;
        for (int i = 0; i < ph * HL_W4W_STAGGER; ++i) __builtin_amdgcn_s_sleep(32);
    }
    // accumulators [frequency f of the wave's 3x3 block][32-channel half cb]: tile f*2+cb; tiles 0..15 = the accumulator file, 16 / 17 here
    f32x16 accv[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[f][r] = 0.f;
    acc_zero();

    auto run = [&](auto wc) {
        constexpr int W = decltype(wc)::value, FR = W >> 1, FC = W & 1, NPW = (NP - W + 3) / 4;
        // ---- patch: per-lane source offsets (fixed for the whole K walk), exactly k_conv_wino4's LDS image ----
        unsigned pv[NPW];
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int c = (W + 4 * j) * 64 + lane;
            int row, col, h;
            if (BLK) {   // eight lanes = the four pixels 4g..4g+3 of a row = 128 contiguous bytes, order inside the group swizzled
                row = c / 72;
                const int g = (c - row * 72) >> 3, kk = (c & 7) ^ ((((g >> 1) & 1) << 2) | ((row >> 2) & 3));
                col = 4 * g + (kk >> 1); h = kk & 1;
            } else {     // lanes 2i, 2i+1 fetch the two halves of one pixel
                const int pixp = c >> 1, q = pixp % PRW;
                row = pixp / PRW; h = (c & 1) ^ ((row >> 2) & 1);
                col = (q % 9) * 4 + q / 9;
            }
            const int y = y0 - 1 + row, x = x0 - 1 + col;
            const bool ok = c < P_REAL && col < 34 && y >= 0 && y < Hv && x >= 0 && x < Wv;
            const int ys = UPS ? y >> 1 : y, xs = UPS ? x >> 1 : x;
            pv[j] = ok ? (unsigned)((img * p.Hin + ys) * p.Win + xs) * pitch4 + h * 16 : OOB;
        }
        f32x4 pp[NPW];                                               // staging registers of the patch pieces in flight
        auto load_pieces = [&](int soffA) {
#pragma unroll
            for (int j = 0; j < NPW; ++j) pp[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, pv[j], soffA, 0));
        };
        // LDS float offset of this lane's 16 bytes inside a piece; the tail lanes of the ragged last piece (P_REAL is not a multiple of 64)
        // are pointed at a dump slot behind the two stages instead of being switched off - no EXEC games between the MFMAs
        constexpr int LASTJ = NPW - 1, RAGGED = (W + 4 * LASTJ + 1) * 64 > P_REAL;
        const int lane_off = lane * 4;
        const int last_off = (!RAGGED || lane < P_REAL - (W + 4 * LASTJ) * 64) ? (W + 4 * LASTJ) * 256 + lane * 4 : 2 * P_F + lane * 4 - (RAGGED ? 0 : 0);
        auto store_piece = [&](int stage, auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j < NPW) {
                if constexpr (j == LASTJ && RAGGED) {
                    // (the dump slot must not move with the stage: lanes that keep out write to 2 P_F + lane*4 for either stage)
                    const int off = (lane < P_REAL - (W + 4 * LASTJ) * 64) ? stage * P_F + last_off : last_off;
                    *reinterpret_cast<f32x4 *>(lds + off) = pp[j];
                } else *reinterpret_cast<f32x4 *>(lds + stage * P_F + (W + 4 * j) * 256 + lane_off) = pp[j];
            }
        };
        // ---- window reads of the transform ----
        const int T = lane & 31, ty = T >> 3, tx = T & 7;
        const float *pread = lds + (((4 * ty + FR) * PRW + tx) * 2 + (half ^ (ty & 1))) * 4;
        int Av[2][2];
#pragma unroll
        for (int cq = 0; cq < 2; ++cq)
#pragma unroll
            for (int rq = 0; rq < 2; ++rq)
                Av[cq][rq] = ((4 * ty + FR) * 9 + tx) * 128 + ((half ^ (((((tx + cq) >> 1) & 1) << 2) | ((ty + rq) & 3))) << 4);
        const int flip = (half ^ (ty & 1)) ? -4 : 4;                  // rows 4, 5 of the window: the other half-slot of the pixel
        auto read_row = [&](int stage, auto kc, auto rc) -> f32x4 {
            constexpr int k = decltype(kc)::value, rr = decltype(rc)::value;
            constexpr int ord0[5] = {4, 2, 0, 3, 1}, ord1[5] = {3, 1, 4, 2, 0};
            constexpr int c = FC + (FC == 0 ? ord0[k] : ord1[k]);
            if (BLK) {
                const char *sb = reinterpret_cast<const char *>(lds + stage * P_F);
                return *reinterpret_cast<const f32x4 *>(sb + ((Av[c >> 2][(FR + rr) >> 2] ^ ((c & 3) << 5)) + (rr * 9 + (c >> 2)) * 128));
            }
            constexpr int coff = ((c & 3) * 9 + (c >> 2)) * 8;
            return *reinterpret_cast<const f32x4 *>(pread + stage * P_F + coff + rr * PRW * 8 + (((FR + rr) >> 2) & 1) * flip);
        };
        Tr tr;
        // transform pieces of the patch in `stage`: one window read, or the whole arithmetic of column step k (+ the final combination)
        auto tr_read = [&](int stage, auto kc, auto rc) { tr.x[decltype(rc)::value] = read_row(stage, kc, rc); };
        auto tr_burst = [&](f32x4 (&Vn)[9], auto kc) {
            constexpr int k = decltype(kc)::value;
            [&]<int... I>(std::integer_sequence<int, I...>) { (op_fwd<FR, I>(tr.x, tr), ...); }(std::make_integer_sequence<int, 6>{});
            op_assign<FC, k>(tr);
            [&]<int... I>(std::integer_sequence<int, I...>) { (op_acc<FC, k, I>(tr), ...); }(std::make_integer_sequence<int, n_acc(FC, k)>{});
            if constexpr (k == 4) [&]<int... I>(std::integer_sequence<int, I...>) { (op_fin<FC, I>(tr, Vn), ...); }(std::make_integer_sequence<int, 6>{});
        };
        // ---- weights: 16 bytes per lane and (frequency, channel half), straight from L2 into a ring of six frequency pairs ----
        const unsigned uvp = (unsigned)lane * 16u;
        const int cbstep = nkt * 36 * 1024;                           // second 32-channel half of the workgroup's 64: the next packed block
        const int ubase = ((2 * nb * nkt + kt0) * 36 + W * 9) * 1024; // (k-tile kt0, frequency 0, half 0) of this wave
#ifndef HL_W4W_RING
#define HL_W4W_RING 3   // frequency pairs in the weight ring (3 or 6): prefetch distance 24 or 48 MFMAs (measured equal: 120.8 vs 120.2 denoise-steps/s)
#endif
        constexpr int RING = HL_W4W_RING;
        static_assert(RING == 3 || RING == 6, "weight ring");
        f32x4 U[2 * RING];                                            // global pair g = 9 t + f lives in U[(g % RING) * 2 + cb] (RING 6: the phase alternates with t & 1)
        auto load_u = [&](int soffU, auto fc, auto slotc) {           // frequency pair f of the k-tile at soffU -> ring slot
            constexpr int f = decltype(fc)::value, sl = decltype(slotc)::value;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                U[sl * 2 + cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, uvp, soffU + f * 1024 + cb * cbstep, 0));
        };

#ifndef HL_W4W_ROTATE
#define HL_W4W_ROTATE 0
#endif
        // HL_W4W_ROTATE = 1 (tried, measured, off): every workgroup walks K from its own starting k-tile and wraps, so that the 32 CUs of
        // an XCD - which run in step - do not ask the L2 for the SAME weight fragments at the same time (the SQ counters show 10 % of the
        // wave cycles parked at s_waitcnt although the loads are issued 24 MFMAs ahead).  Result: 15 % SLOWER (1 712 -> 1 970 us at 768
        // input channels, 519 -> 564 us at 192): the lock-step is what makes one CU's miss everybody else's hit.
        const int rot = HL_W4W_ROTATE ? (int)(((unsigned)tb * 7u + (unsigned)nb * 3u) % (unsigned)max(ntiles, 1)) : 0;
        auto tmap = [&](int t) { const int x = min(t, ntiles - 1) + rot; return x >= ntiles ? x - ntiles : x; };   // t-th k-tile this workgroup takes
        f32x4 V0[9], V1[9];
        if (ntiles > 0) {
            // prologue: patches 0 and 1 into the two stages, the first six weight pairs, the transform of patch 0 (nothing to overlap it with)
            load_pieces((kt0 + tmap(0)) * kstep);
            [&]<int... J>(std::integer_sequence<int, J...>) { (store_piece(0, std::integral_constant<int, J>{}), ...); }(std::make_integer_sequence<int, 6>{});
            load_pieces((kt0 + tmap(1)) * kstep);
            [&]<int... J>(std::integer_sequence<int, J...>) { (store_piece(1, std::integral_constant<int, J>{}), ...); }(std::make_integer_sequence<int, 6>{});
            [&]<int... F>(std::integer_sequence<int, F...>) { (load_u(ubase + tmap(0) * (36 * 1024), std::integral_constant<int, F>{}, std::integral_constant<int, F>{}), ...); }(std::make_integer_sequence<int, RING>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            [&]<int... K>(std::integer_sequence<int, K...>) {
                ([&] {
                    constexpr int Kc = K;   // (a pack name inside the inner fold would be expanded in lockstep with R)
                    [&]<int... R>(std::integer_sequence<int, R...>) { (tr_read(0, std::integral_constant<int, Kc>{}, std::integral_constant<int, R>{}), ...); }(std::make_integer_sequence<int, 5>{});
                    tr_burst(V0, std::integral_constant<int, Kc>{});
                }(), ...);
            }(std::make_integer_sequence<int, 5>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                             // every wave has read stage 0: it may be refilled
            asm volatile("" ::: "memory");
        }
        // One k-tile = 72 gaps.  MFMA order inside a frequency pair f: (cb0,s0) (cb1,s0) (cb0,s1) ... - consecutive MFMAs never share an
        // accumulator.  TP = t & 1 fixes the ring phase (9 pairs per k-tile on a ring of 6).  Nothing in the body depends on t otherwise:
        // past the end the loads re-read the last k-tile and the transform chews on a stale stage.
        auto body = [&](auto sc, f32x4 (&Vc)[9], f32x4 (&Vn)[9], int t) {
            constexpr int S = decltype(sc)::value;                    // = t & 1: stage of patch t
            const int tl0 = tmap(t), tl1 = tmap(t + 1), tl2 = tmap(t + 2);
            const int soffU0 = (HL_W4W_ABL & 64) ? ubase : ubase + tl0 * (36 * 1024), soffU1 = (HL_W4W_ABL & 64) ? ubase : ubase + tl1 * (36 * 1024);   // (64: hot weights)
            const int soffA2 = (HL_W4W_ABL & 128) ? kt0 * kstep : (kt0 + tl2) * kstep;                                                                   // (128: hot patch)
            [&]<int... J>(std::integer_sequence<int, J...>) {
                ([&] {
                    constexpr int Jc = J, f = J >> 3, idx = J & 7, s = idx >> 1, cb = idx & 1;
                    constexpr int g = 9 * S + f, us = (g % RING) * 2;     // global pair index (mod 18), its ring slot
                    if constexpr (!(HL_W4W_ABL & 16)) mfma_tile<f * 2 + cb>(accv[(f * 2 + cb) & 1], Vc[f][s], U[us + cb][s]);
                    if constexpr ((HL_W4W_ABL & 32) != 0) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!(HL_W4W_ABL & 2)) {
                        // window reads of column step k: one per gap, after the burst that consumed the column buffer
                        [&]<int... K>(std::integer_sequence<int, K...>) {
                            ([&] {
                                if constexpr (Jc >= read_gap(K, 0) && Jc < read_gap(K, 0) + 5)
                                    tr_read(S ^ 1, std::integral_constant<int, K>{}, std::integral_constant<int, Jc - read_gap(K, 0)>{});
                                if constexpr (Jc == GAP_BURST0 + K * GAP_BURST_STEP) tr_burst(Vn, std::integral_constant<int, K>{});
                            }(), ...);
                        }(std::make_integer_sequence<int, 5>{});
                    }
                    // The MFMA reads its A / B operands while it runs, and the compiler - to which the asm statement is opaque - hands operand
                    // registers that die at an MFMA to the instructions right behind it (a burst overwrote the weight fragment of the MFMA in
                    // front of it: wrong sums).  The operands of this gap's MFMA and of the one before stay alive to the end of the gap.
                    if constexpr (idx >= 1) {   // (idx 0: the previous MFMA's fragment has already been handed to a load that lands much later)
                        asm volatile("" ::"v"(Vc[f]), "v"(U[us + ((idx - 1) & 1)]));
                    } else if constexpr (J >= 1) asm volatile("" ::"v"(Vc[f - 1]));
                    asm volatile("" ::"v"(Vc[f]), "v"(U[us + cb]));
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (idx == 7 && !(HL_W4W_ABL & 4)) {   // pair f is consumed: its ring slot takes pair f+3 (of this k-tile or the next)
                        // (its slot takes pair g + RING: RING pairs = 8 RING MFMAs ahead; patch loads that sit in front of it in the in-order
                        //  return queue then have that long to land before they can stall a weight wait)
                        if constexpr (f + RING < 9) load_u(soffU0, std::integral_constant<int, f + RING>{}, std::integral_constant<int, g % RING>{});
                        else load_u(soffU1, std::integral_constant<int, f + RING - 9>{}, std::integral_constant<int, g % RING>{});
                    }
                    if constexpr (!(HL_W4W_ABL & 8)) {
                        if constexpr (Jc == GAP_PATCH_LOAD) load_pieces(soffA2);
                        if constexpr (Jc >= GAP_PATCH_STORE && Jc < GAP_PATCH_STORE + 6) store_piece(S, std::integral_constant<int, Jc - GAP_PATCH_STORE>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }(), ...);
            }(std::make_integer_sequence<int, 72>{});
            // patch t+2 is in LDS (this wave's share); all waves are done with stage S^1
            if constexpr (!(HL_W4W_ABL & 1)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            asm volatile("" ::: "memory");
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        int t = 0;
        for (; t + 1 < ntiles; t += 2) { body(S0{}, V0, V1, t); body(S1{}, V1, V0, t + 1); }
        if (t < ntiles) body(S0{}, V0, V1, t);
    };
    switch (wave) {
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        default: run(std::integral_constant<int, 3>{}); break;
    }
    // stray patch pieces must not land in the exchange buffer; the last MFMAs have left the pipe before their results are read
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");

    // ---- output transform: the 36 frequencies of a (tile, channel) meet in LDS, two rounds of 16 tiles x 64 channels: [freq][tile][cout]
    // (144 KB - the workgroup has the CU's LDS to itself).  A thread finishes one tile for FOUR neighbouring channels (ds_read_b128, 16-byte
    // stores: 16 lanes cover the 256 bytes of a pixel) - the epilogue is bound by the number of store instructions, not by their bytes.
    const int fbase = (3 * (wave >> 1)) * 6 + 3 * (wave & 1);
    const int mloc = tid >> 4, nq = (tid & 15) * 4, n = n0 + nq;
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && !p.partial) bs = *reinterpret_cast<const f32x4 *>(p.bias + n);
    const long hw = (long)Hv * Wv;
    auto w4_out4 = [](f32x4 m0, f32x4 m1, f32x4 m2, f32x4 m3, f32x4 m4, f32x4 m5, f32x4 (&y)[4]) {
        const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
        y[0] = (m0 + s1) + s2;
        y[1] = __builtin_elementwise_fma((f32x4)(W4_B), d2, W4_A * d1);
        y[2] = __builtin_elementwise_fma((f32x4)(W4_B * W4_B), s2, (W4_A * W4_A) * s1);
        y[3] = __builtin_elementwise_fma((f32x4)(W4_B * W4_B * W4_B), d2, __builtin_elementwise_fma((f32x4)(W4_A * W4_A * W4_A), d1, m5));
    };
    f32x4 st_sm[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, st_sq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // GroupNorm sums of out / out2
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        __syncthreads();
        if (!((HL_W4W_ABL & 512) && p.Hin != 12345))
        [&]<int... I>(std::integer_sequence<int, I...>) {   // I = (f, cb, rr): accumulator register 8q + rr of tile f*2+cb holds tile 16q + (rr&3) + 8(rr>>2) + 4 half
            ([&] {
                constexpr int f = I / 16, cb = (I >> 3) & 1, rr = I & 7;
                const int F = fbase + (f / 3) * 6 + (f % 3);
                const float v = q == 0 ? acc_get<f * 2 + cb, rr>(accv) : acc_get<f * 2 + cb, 8 + rr>(accv);
                lds[(F * 16 + (rr & 3) + 8 * (rr >> 2) + 4 * half) * 64 + cb * 32 + (lane & 31)] = v;
            }(), ...);
        }(std::make_integer_sequence<int, 144>{});
        __syncthreads();
        const float *zz = lds + mloc * 64 + nq;
        f32x4 z[4][6];                                                // rows of A^T applied: z[p][j]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x4 col[4];
            w4_out4(*reinterpret_cast<const f32x4 *>(zz + (0 * 6 + j) * 1024), *reinterpret_cast<const f32x4 *>(zz + (1 * 6 + j) * 1024),
                    *reinterpret_cast<const f32x4 *>(zz + (2 * 6 + j) * 1024), *reinterpret_cast<const f32x4 *>(zz + (3 * 6 + j) * 1024),
                    *reinterpret_cast<const f32x4 *>(zz + (4 * 6 + j) * 1024), *reinterpret_cast<const f32x4 *>(zz + (5 * 6 + j) * 1024), col);
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) z[pr][j] = col[pr];
        }
        const int Tg = q * 16 + mloc, oy = y0 + 4 * (Tg >> 3), ox = x0 + 4 * (Tg & 7);
        const long m0 = ((long)img * Hv + oy) * Wv + ox;              // pixel (pr, qc) of the tile: m0 + pr*Wv + qc
        f32x4 v[16];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            f32x4 row[4];
            w4_out4(z[pr][0], z[pr][1], z[pr][2], z[pr][3], z[pr][4], z[pr][5], row);
#pragma unroll
            for (int qc = 0; qc < 4; ++qc) v[pr * 4 + qc] = row[qc] + bs;
        }
        if (p.partial) {   // split-K: the output transform is linear, so slabs are summed in the output domain by k_splitk_finish
            float *dst = p.partial + (long)blockIdx.z * p.M * p.Cout + m0 * p.Cout + n;
#pragma unroll
            for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x4 *>(dst + (long)((k >> 2) * Wv + (k & 3)) * p.Cout) = v[k];
            continue;
        }
        if ((HL_W4W_ABL & 256) && p.Hin != 12345) continue;
        if (p.res) {
            const float *rp = p.res + m0 * p.res_pitch + n;
            f32x4 rr[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) rr[k] = *reinterpret_cast<const f32x4 *>(rp + (long)((k >> 2) * Wv + (k & 3)) * p.res_pitch);
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] += rr[k];
        }
        f32x4 v2[16];
        if (p.out2) {
            const float *rp = p.res2 + m0 * p.res2_pitch + n;
#pragma unroll
            for (int k = 0; k < 16; ++k) v2[k] = *reinterpret_cast<const f32x4 *>(rp + (long)((k >> 2) * Wv + (k & 3)) * p.res2_pitch);
#pragma unroll
            for (int k = 0; k < 16; ++k) v2[k] += v[k];
        }
        // GroupNorm statistics: the wave's four tiles of the round = 64 pixels of one image; the sums of both rounds stay in registers
        auto stats = [&](f32x4 &sm_t, f32x4 &sq_t, const f32x4(&vv)[16]) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { sm_t += vv[k]; sq_t += vv[k] * vv[k]; }
        };
        if (p.st1) stats(st_sm[0], st_sq[0], v);
        if (p.st2) stats(st_sm[1], st_sq[1], v2);
        if (p.out_nchw) {
            float *op = p.out + ((long)img * p.Cout + n) * hw + (m0 - (long)img * hw);
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) op[c * hw + (k >> 2) * Wv + (k & 3)] = v[k][c];
        } else {
            float *op = p.out + m0 * p.out_pitch + n;
#pragma unroll
            for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x4 *>(op + (long)((k >> 2) * Wv + (k & 3)) * p.out_pitch) = v[k];
        }
        if (p.out2) {
            float *op = p.out2 + m0 * p.out2_pitch + n;
#pragma unroll
            for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x4 *>(op + (long)((k >> 2) * Wv + (k & 3)) * p.out2_pitch) = v2[k];
        }
    }
    if (p.st1 || p.st2) {   // the four waves' sums meet in LDS (the exchange buffer is free): ONE atomic pair per channel and workgroup
        __syncthreads();
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
            if (!(w2 ? p.st2 : p.st1)) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                st_sm[w2][i] += __shfl_xor(st_sm[w2][i], 16); st_sm[w2][i] += __shfl_xor(st_sm[w2][i], 32);
                st_sq[w2][i] += __shfl_xor(st_sq[w2][i], 16); st_sq[w2][i] += __shfl_xor(st_sq[w2][i], 32);
            }
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wg_stat_put<64>(lds + w2 * 512, wave, nq + i, st_sm[w2][i], st_sq[w2][i]);
            }
        }
        __syncthreads();
        if (p.st1) wg_group_flush<64, 4>(lds, reinterpret_cast<unsigned long long *>(lds + 1024), p.st1, p.N, img, p.Cout, n0, p.st1_c0, p.st1_cg, hw, tid);
        if (p.st2) wg_group_flush<64, 4>(lds + 512, reinterpret_cast<unsigned long long *>(lds + 1024), p.st2, p.N, img, p.Cout, n0, p.st2_c0, p.st2_cg, hw, tid);
    }
#endif
}

}  // namespace

size_t conv_wino4w_lds_bytes() { return (size_t)36 * 16 * 64 * sizeof(float); }   // 144 KB: the output exchange (the two patch stages + the dump slot need 41.5 KB)

int conv_wino4w_launch(const ConvK &p, int ups, int blk, int splits, hipStream_t st) {
    HL_REQUIRE(p.Cout % 64 == 0 && p.Cin % 8 == 0 && p.w_wino, "k_conv_wino4w: bad layer");
    HL_REQUIRE(!(ups && blk), "k_conv_wino4w: the upsampling convolution reads a raw NHWC tensor");
    const dim3 grid((unsigned)(p.n_mtiles * p.n_nblocks), 1, (unsigned)splits);
    const size_t sh = conv_wino4w_lds_bytes();
    static const bool attr_ok = [] {   // 144 KB of dynamic LDS: above the default cap
        const int b = (int)conv_wino4w_lds_bytes();
        return hipFuncSetAttribute((const void *)k_conv_wino4w<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, b) == hipSuccess &&
               hipFuncSetAttribute((const void *)k_conv_wino4w<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, b) == hipSuccess &&
               hipFuncSetAttribute((const void *)k_conv_wino4w<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, b) == hipSuccess;
    }();
    HL_REQUIRE(attr_ok, "k_conv_wino4w: cannot raise the dynamic LDS limit to %zu bytes", sh);
    if (ups) hipLaunchKernelGGL((k_conv_wino4w<true, false>), grid, dim3(256), sh, st, p);
    else if (blk) hipLaunchKernelGGL((k_conv_wino4w<false, true>), grid, dim3(256), sh, st, p);
    else hipLaunchKernelGGL((k_conv_wino4w<false, false>), grid, dim3(256), sh, st, p);
    return check_launch("k_conv_wino4w");
}

}  // namespace hl
