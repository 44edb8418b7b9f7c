// k_conv_h16: the 3x3 / stride-1 convolutions of the UNet with 16-bit operands (fp16 or bf16) and fp32 accumulation on
// v_mfma_f32_32x32x16_{f16,bf16} - the arithmetic the reference's training scripts select (train_util.py:214 torch.autocast under
// --use_amp True; fp16 there) and, for fp16, the operand precision the reference's own convolutions have on its hardware (TF32: the
// same 10 explicit significand bits).  Opt-in (HL_CONV_BF16 / HL_CONV_FP16), never the default: the fp32 kernels stay the product path.
//
// Direct implicit GEMM, one workgroup = 16x16 output pixels x 192 output channels, ONE wave per SIMD (the 512-entry register budget):
//   * wave (wm, wn) owns 128 pixels (rows 8wm..8wm+7 of the tile) x 96 channels = 4 x 3 accumulator tiles of 32x32 (192 registers,
//     which the compiler keeps in the accumulator file);
//   * K is walked in chunks of 32 input channels: the chunk's 18x18 input patch (zero padding = out-of-range buffer loads) is fetched
//     as fp32 into registers one chunk ahead, rounded to 16 bits (nearest even) and written to LDS as [pixel][32 channels] (64 bytes per
//     pixel, rows of 20 pixels, the 16-byte quarter index XORed with bits 2-3 of the pixel's x: the slot (address / 16 mod 16) of a
//     quarter is then a bijection of x mod 16, and each of the four 16-lane groups a ds_read_b128 is served in holds 16 consecutive x
//     - conflict-free for every tap), two stages; the nine taps of the chunk read the SAME patch at shifted pixel addresses;
//   * the weights never touch LDS: they are packed in MFMA-fragment order (one 16-byte load per lane and fragment) and go from L2 into a
//     register ring of six k-steps (3 fragments each) - the CUs of an XCD walk K in step, so every line is one miss and 31 hits;
//   * per k-step (16 channels of one tap): 4 ds_read_b128 + 3 buffer_load_dwordx4 for 12 MFMAs of 32 cycles.
// Epilogue: two rounds of 128 pixels x 192 channels through LDS (padded rows), then bias + residual (+ the control tower's second output) and
// 16-byte NHWC stores, GroupNorm statistics per (tile, round).
// Variants in this file: the nearest-x2 upsample in front of the convolution (a source-address shift of the patch loads); a 16-bit activation
// image written by k_gn_apply_h16 for layers behind a GroupNorm (ConvK::in16: no rounding in the staging); split-K over chunk slabs for
// layers with few tiles (blockIdx.z, raw sums to the split-K workspace, k_splitk_finish adds them in a fixed order); k_conv1_h16 for the 1x1
// layers (256 consecutive pixels per workgroup, chunks of 96 channels).  The weight gradients live in hl_unet_train.hip (k_conv_wgrad_h16).
#include "hl_unet_kernels.h"
#include "hl_stats.h"

#include <type_traits>
#include <utility>

namespace hl {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <bool F16>
__device__ __forceinline__ unsigned pack2(float a, float b) {   // two 16-bit values (round to nearest even), a in the low half
    if constexpr (F16) {
        const f16x2 h = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, h);
    } else {
        const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
        const unsigned ra = ua + 0x7fffu + ((ua >> 16) & 1u), rb = ub + 0x7fffu + ((ub >> 16) & 1u);
        return __builtin_amdgcn_perm(rb, ra, 0x07060302);
    }
}

template <bool F16>
__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int H16_PW = 18, H16_NPIX = H16_PW * H16_PW;       // patch of a 16x16 tile
constexpr int H16_LP = 20;                                    // LDS row pitch of the patch in pixels (multiple of 4: the slot of a pixel depends on its x only)
constexpr int H16_STAGE = H16_PW * H16_LP * 64;              // bytes per patch stage
#ifndef H16_RING
#define H16_RING 6                                         // k-steps of weights in flight (divides 18)
#endif
#ifndef H16_ABL   // timing ablations (wrong results): 1 no weight loads in the loop, 2 no patch loads / staging in the loop, 4 no A-fragment reads, 8 no MFMA
#define H16_ABL 0
#endif
#ifndef H16_ST0
#define H16_ST0 12                                        // k-step behind which the next chunk's patch goes to LDS (6 units, one per step)
#endif
constexpr int H16_EP = 196;                                  // epilogue row pitch in floats (192 + 4: rows 4 apart are 16 banks apart)

// Epilogue shared by the 3x3 and the 1x1 kernel: the wave tiles (128 pixel slots x 96 channels per wave) go through LDS in two rounds of
// (fragments 2q, 2q+1 of every wave) = 128 pixel slots x 192 channels; pix(q, slot) = the output pixel of slot (wave half wm = slot >> 6,
// fragment 2q + ((slot >> 5) & 1), row slot & 31); bias, residual, 16-byte NHWC stores, GroupNorm statistics per (tile, round).
template <class PixFn>
__device__ __forceinline__ void h16_epilogue(const ConvK &p, char *lds, const f32x16 (&acc)[4][3], int tid, int lane, int wm, int wn, int n0, long tile,
                                             PixFn pix) {
    // ---- epilogue: two rounds of (fragments 2q, 2q+1 of every wave) = 128 pixels x 192 channels through LDS
    float *ep = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q) __syncthreads();
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int nf = 0; nf < 3; ++nf)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = (i >> 2) * 8 + (lane >> 5) * 4 + (i & 3);            // row of the 32x32 tile held by acc[..][i]
                    const int prow = wm * 64 + m2 * 32 + r;                            // pixel slot of the round: wave-major, fragment, row
                    ep[prow * H16_EP + wn * 96 + nf * 32 + (lane & 31)] = acc[2 * q + m2][nf][i];
                }
        __syncthreads();
        // thread (pr0 = tid / 48 < 5, cq = tid % 48) finishes channels 4cq..4cq+3 of pixel slots pr0, pr0 + 5, ... (<= 26 of the round's 128),
        // in batches of 9 / 9 / 8: the residual loads of a batch are all in flight before its first store (on this ISA stores count on
        // vmcnt too: a load waited for behind a store waits for the store).  One channel quad per thread = GroupNorm sums without shuffles.
        const int pr0 = tid / 48, cq = (tid - pr0 * 48) * 4;
        f32x4 sm = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f}, sm2 = {0.f, 0.f, 0.f, 0.f}, sq2 = {0.f, 0.f, 0.f, 0.f};
        f32x4 bs = {0.f, 0.f, 0.f, 0.f}, ws = {1.f, 1.f, 1.f, 1.f};
        if (p.bias && pr0 < 5) bs = *reinterpret_cast<const f32x4 *>(p.bias + n0 + cq);
        if (p.wsc && pr0 < 5) ws = *reinterpret_cast<const f32x4 *>(p.wsc + n0 + cq);      // fp16x2 planes: the inverse power-of-two scales of these four channels' weights
        auto finish = [&](auto kc0, auto nbc) {   // pixel slots pr0 + 5 (K0 .. K0 + NB - 1): every load of the batch in flight before its first store
            constexpr int K0 = decltype(kc0)::value, NB = decltype(nbc)::value;
            long mm[NB];
            f32x4 v[NB], rr[NB], r2[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int pc = pr0 + 5 * (K0 + k);
                mm[k] = pix(q, pc);
                if (p.res && !p.partial) rr[k] = *reinterpret_cast<const f32x4 *>(p.res + mm[k] * p.res_pitch + n0 + cq);
                if (p.out2 && !p.partial) r2[k] = *reinterpret_cast<const f32x4 *>(p.res2 + mm[k] * p.res2_pitch + n0 + cq);
                v[k] = *reinterpret_cast<const f32x4 *>(ep + pc * H16_EP + cq) * ws;
            }
            if (p.partial) {   // split-K slab: raw sums, [z][pixel][Cout]
#pragma unroll
                for (int k = 0; k < NB; ++k) *reinterpret_cast<f32x4 *>(p.partial + ((long)blockIdx.z * p.M + mm[k]) * p.Cout + n0 + cq) = v[k];
                return;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                v[k] += bs;
                if (p.res) v[k] += rr[k];
                sm += v[k]; sq += v[k] * v[k];
                *reinterpret_cast<f32x4 *>(p.out + mm[k] * p.out_pitch + n0 + cq) = v[k];
                if (p.out2) {                                   // second output out2 = out + res2 (the control tower's zero convolution: unet.py:598-602)
                    r2[k] += v[k];
                    sm2 += r2[k]; sq2 += r2[k] * r2[k];
                    *reinterpret_cast<f32x4 *>(p.out2 + mm[k] * p.out2_pitch + n0 + cq) = r2[k];
                }
            }
        };
        if (pr0 < 5) {
            finish(std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
            finish(std::integral_constant<int, 9>{}, std::integral_constant<int, 8>{});
            finish(std::integral_constant<int, 17>{}, std::integral_constant<int, 8>{});
            if (pr0 < 3) finish(std::integral_constant<int, 25>{}, std::integral_constant<int, 1>{});      // slots 125, 126, 127
        }
        auto put_stats = [&](float *st, const f32x4 a_sm, const f32x4 a_sq) {   // (tile, round) = 128 pixels of one image; the five pixel groups meet in LDS
            __syncthreads();
            if (pr0 < 5) {
                *reinterpret_cast<f32x4 *>(ep + (pr0 * 48 * 2 + (cq >> 2) * 2) * 4) = a_sm;
                *reinterpret_cast<f32x4 *>(ep + (pr0 * 48 * 2 + (cq >> 2) * 2 + 1) * 4) = a_sq;
            }
            __syncthreads();
            if (tid < 48) {
                f32x4 a = *reinterpret_cast<const f32x4 *>(ep + (tid * 2) * 4), b = *reinterpret_cast<const f32x4 *>(ep + (tid * 2 + 1) * 4);
#pragma unroll
                for (int g = 1; g < 5; ++g) {
                    a += *reinterpret_cast<const f32x4 *>(ep + (g * 48 * 2 + tid * 2) * 4);
                    b += *reinterpret_cast<const f32x4 *>(ep + (g * 48 * 2 + tid * 2 + 1) * 4);
                }
                const long img = (tile * 256) / ((long)p.Hout * p.Wout);   // (a tile = 256 pixels of one image)
                // the thread's four consecutive channels: one atomic pair per group they fall in (atomics on one word serialise)
                const int c0_ = st == p.st1 ? p.st1_c0 : p.st2_c0, cg_ = st == p.st1 ? p.st1_cg : p.st2_cg;
                int g_run = (c0_ + n0 + tid * 4) / cg_;
                float s_run = a[0], q_run = b[0];
#pragma unroll
                for (int i = 1; i < 4; ++i) {
                    const int g_i = (c0_ + n0 + tid * 4 + i) / cg_;
                    if (g_i != g_run) { stat_add(st, p.N, img, g_run, (long)p.Hout * p.Wout, s_run, q_run); g_run = g_i; s_run = 0.f; q_run = 0.f; }
                    s_run += a[i]; q_run += b[i];
                }
                stat_add(st, p.N, img, g_run, (long)p.Hout * p.Wout, s_run, q_run);
            }
        };
        if (p.st1 && !p.partial) put_stats(p.st1, sm, sq);      // GroupNorm statistics of the stored tensor(s)
        if (p.st2 && !p.partial) put_stats(p.st2, sm2, sq2);
    }
}

#ifndef HL_ACT_SCALE
#define HL_ACT_SCALE 1      // developer A / B: 0 = no power-of-two activation scale (sx = 1 everywhere)
#endif
struct H2Pair { unsigned p0, p1; };
// a pair of values -> the two packed planes.  Default (round 6, H2_SPLIT_RNE = 2): h0 = the NEAREST fp16 (v_cvt_pk_f16_f32), h1 = the nearest fp16 of the residual
// against h0 (v_fma_mix_f32 reads the packed halves: exact in fp32): |x - h0 - h1| <= 2^-24 |x| while both planes are normal, FOUR instructions per pair
// (hl_split2_rne, hl_common.h) where round 5's truncating split took six, and a value beyond fp16's range becomes inf / NaN (loud) where v_cvt_pkrtz_f16_f32
// saturated at 65504 without a trace.  The staging scales its input so that neither a plane overflows nor the low plane goes subnormal wherever the tensor's
// totals are known (hl_stats.h).  rel-L2 against float64 3.50e-7 / 3.43e-7 / 3.28e-7 (stride-2, K = 864) for H2_SPLIT_RNE = 0 / 1 / 2.  Time: the forward is
// 1 % (B = 1) to 3 % (B = 4) slower than round 5's on the same box, and none of it is instructions - planes that keep all their mantissa bits make the matrix
// pipe issue fewer MFMAs per clock (same reported shader clock; profiles/r06_unet_regression.md, scripts/microbench/mfma_data_power.hip).
// 0: h0 = the value with its low 13 mantissa bits cleared, both conversions truncating (round 5); 1: the same h0, h1 to nearest.
#ifndef H2_SPLIT_RNE
#define H2_SPLIT_RNE 2
#endif
__device__ __forceinline__ H2Pair split_h2(float x, float y) {
    const float hx = __builtin_bit_cast(float, __float_as_uint(x) & 0xffffe000u), hy = __builtin_bit_cast(float, __float_as_uint(y) & 0xffffe000u);
#if H2_SPLIT_RNE == 2
    H2Pair r;
    hl_split2_rne(x, y, r.p0, r.p1);
    return r;
#elif H2_SPLIT_RNE
    return H2Pair{pack2<true>(hx, hy), pack2<true>(x - hx, y - hy)};
#else
    return H2Pair{__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(hx, hy)), __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x - hx, y - hy))};
#endif
}


// NPL = operand planes: 1 = 16-bit operands (the opt-in modes), 2 = fp16x2 (round 5: the DEFAULT mode's 3x3 layers where this kernel beats the fp32 Winograd
// kernels): two fp16 planes per operand - activations h0 = the value with its low 13 mantissa bits cleared, h1 = the truncated residual (2^-20), weights
// nearest-even planes (2^-22) - and the three partial products h1 w0 + h0 w1 + h0 w0 per k-step, fp32 accumulation: a DIRECT convolution, no transform
// in front of the products, so its error is the fp32 direct kernel's class (and below F(4x4,3x3)'s).  Planes: LDS stage = [plane][patch], weights
// [..][k-half][plane][wn][fragment][lane][8], a two-plane activation image (k_gn_apply_*: ConvK::in16 = 2) is [plane][pixel][C].
template <bool F16, int NPL>
__global__ __launch_bounds__(256, 1) void k_conv_h16(const ConvK p) {
#if __HIP_DEVICE_COMPILE__
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NU = H16_NPIX * 4, NUT = (NU + 255) / 256;      // staging units (pixel, 8-channel group) and units per thread
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;       // contiguous runs of (tile, channel block) per XCD
    const int tb = __builtin_amdgcn_readfirstlane(wi / p.n_nblocks), nb = wi - tb * p.n_nblocks;   // (wave-uniform: the weight loads take them as scalar offsets)
    const int bw = p.Wout >> 4, bh = p.Hout >> 4;
    const int img = tb / (bw * bh), brem = tb - img * (bw * bh);
    const int y0 = (brem / bw) * 16, x0 = (brem - (brem / bw) * bw) * 16;
    const int nch = p.Cin >> 5;
    constexpr int RING = NPL == 2 ? 3 : H16_RING;                            // k-steps of weights in flight (two planes: half as deep, the same registers)
    constexpr int STG = NPL * H16_STAGE;                                      // bytes per stage (all planes)
    const unsigned pitch4 = (unsigned)p.in_pitch * (p.in16 ? 2u : 4u);      // bytes per pixel (16-bit image from the GroupNorm pass, or fp32)
    const unsigned plane_b = (unsigned)((long)p.N * p.Hin * p.Win * p.in_pitch * 2);   // (in16 == 2: the second plane of the image)

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.in, (short)0, (int)((long)p.N * p.Hin * p.Win * pitch4 * (p.in16 == 2 ? 2 : 1)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.w_bf3, (short)0, (int)((long)p.n_nblocks * nch * 18 * 6144 * NPL), 0x00020000);

    // ---- patch staging: unit u = (pixel of the 18x18 patch, group of 8 channels) -> two 16-byte fp32 loads, one 16-byte LDS write ----
    unsigned sv[NUT], sl[NUT];
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
        const int u = tid + 256 * j, pix = u >> 2, grp = u & 3;
        const int py = pix / H16_PW, px = pix - py * H16_PW;
        const int y = y0 - 1 + py, x = x0 - 1 + px;                 // (p.ups: coordinates in the nearest-x2 image, unet.py:77 - source pixel (y >> 1, x >> 1))
        const bool ok = u < NU && y >= 0 && y < p.Hout && x >= 0 && x < p.Wout;
        const int ys = p.ups ? y >> 1 : y, xs = p.ups ? x >> 1 : x;
        sv[j] = ok ? (unsigned)((img * p.Hin + ys) * p.Win + xs) * pitch4 + grp * (p.in16 ? 16 : 32) : OOB;
        sl[j] = u < NU ? (unsigned)((py * H16_LP + px) * 64 + ((grp ^ ((px >> 2) & 3)) << 4)) : (unsigned)(2 * STG + (tid & 63) * 16);   // (no unit: a dump slot behind the stages)
    }
    u32x4 ar[NUT][2];
    auto a_load = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NUT; ++j) {
            const int so = (H16_ABL & 16) ? 0 : chunk * (p.in16 ? 64 : 128);      // (16: the patch from L2-hot addresses)
            ar[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j], so, 0);
            if (!p.in16) ar[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j] == OOB ? OOB : sv[j] + 16, so, 0);
            else if (NPL == 2) ar[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j] == OOB ? OOB : sv[j] + plane_b, so, 0);
        }
    };
    auto a_store = [&](int stage, int j) {
        const f32x4 v0 = __builtin_bit_cast(f32x4, ar[j][0]), v1 = __builtin_bit_cast(f32x4, ar[j][1]);
        char *dst = lds + (sl[j] >= 2u * STG ? 0 : stage * STG) + sl[j];
        if constexpr (NPL == 2) {
            u32x4 h0 = ar[j][0], h1 = ar[j][1];
            if (!p.in16) {
                const H2Pair q0 = split_h2(v0[0], v0[1]), q1 = split_h2(v0[2], v0[3]), q2 = split_h2(v1[0], v1[1]), q3 = split_h2(v1[2], v1[3]);
                h0 = u32x4{q0.p0, q1.p0, q2.p0, q3.p0}; h1 = u32x4{q0.p1, q1.p1, q2.p1, q3.p1};
            }
            *reinterpret_cast<u32x4 *>(dst) = h0;
            if (sl[j] < 2u * STG) *reinterpret_cast<u32x4 *>(dst + H16_STAGE) = h1;
        } else {
            const u32x4 h = p.in16 ? ar[j][0]
                                   : u32x4{pack2<F16>(v0[0], v0[1]), pack2<F16>(v0[2], v0[3]), pack2<F16>(v1[0], v1[1]), pack2<F16>(v1[2], v1[3])};
            *reinterpret_cast<u32x4 *>(dst) = h;
        }
    };

    // ---- A fragments: row r of fragment mf = pixel (8wm + 2mf + (r >> 4), r & 15) of the tile; lane half g holds channels 8g..8g+7 of the k-step
    unsigned aoff[4][3];                                           // per tap column kx (the swizzle depends on x); tap row ky is an immediate offset
    {
        const int r = lane & 31, g = lane >> 5;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int py = 8 * wm + 2 * mf + (r >> 4), px = (r & 15) + kx;
                aoff[mf][kx] = (unsigned)((py * H16_LP + px) * 64 + ((g ^ ((px >> 2) & 3)) << 4));
            }
    }
    // ---- weights: packed [channel block][chunk][tap][k-half][wn][fragment][lane][8] -> one 16-byte load per lane and fragment
    const unsigned wv = (unsigned)lane * 16u;
    const int wbase = nb * nch * 18 * 6144 * NPL + wn * 3072;
    u32x4 ring[RING][NPL][3];
    // split-K (layers with few tiles): blockIdx.z owns the chunks [c0, c1) and stores raw accumulators to p.partial (k_splitk_finish sums
    // the slabs in a fixed order and applies bias / residual / statistics).  (A per-workgroup rotation of the chunk order,
    // tried against L2-channel hot spots - changed nothing and is gone.)
    const int c0 = (int)blockIdx.z * p.kt_per, c1 = min(nch, c0 + p.kt_per);
    auto w_load = [&](int slot_, int chunk, int s18) {
        const int so = wbase + (chunk * 18 + s18) * 6144 * NPL;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) ring[slot_][pl][nf] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wv, so + pl * 6144 + nf * 1024, 0);
    };

    f32x16 acc[4][3];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;

    // ---- prologue: the first chunk staged, the first H16_RING k-steps of weights in flight
    a_load(c0);
#pragma unroll
    for (int s = 0; s < RING; ++s) w_load(s, c0 + s / 18 < nch ? c0 + s / 18 : c0, s % 18);
#pragma unroll
    for (int j = 0; j < NUT; ++j) a_store(0, j);
    __syncthreads();

    u32x4 af[2][NPL][4];                                           // A fragments of the current / next k-step (per plane)
    auto a_read = [&](const char *st, int tap, int k2, u32x4 (&dst)[NPL][4]) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) dst[pl][mf] = *reinterpret_cast<const u32x4 *>(st + pl * H16_STAGE + (tap / 3) * (H16_LP * 64) + (aoff[mf][tap % 3] ^ (k2 ? 32u : 0u)));
    };
    for (int c = c0; c < c1; ++c) {
        const char *st = lds + ((c - c0) & 1) * STG;
        const int cc = c;                                                // this chunk / the next one (past the end: a valid chunk, never used)
        const int ccn = cc + 1 < nch ? cc + 1 : 0;
        if (!(H16_ABL & 2) && !(H16_ABL & 64)) a_load(ccn);              // (last chunk: a valid chunk again, staged and never read)
        a_read(st, 0, 0, af[0]);
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... S>(std::integer_sequence<int, S...>) {
            ([&] {
                constexpr int rs = S % RING, cur = S & 1;
                if constexpr (S + 1 < 18 && !(H16_ABL & 4)) a_read(st, (S + 1) >> 1, (S + 1) & 1, af[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);                 // the next step's fragments are on their way before this step's MFMAs
#ifdef H2_ORDER_PRODUCT_OUTER
                if constexpr (NPL == 2 && !(H16_ABL & 8)) {
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int nf = 0; nf < 3; ++nf)
#pragma unroll
                            for (int mf = 0; mf < 4; ++mf)
                                acc[mf][nf] = mma<true>(af[cur][pr == 0 ? 1 : 0][mf], ring[rs][pr == 1 ? 1 : 0][nf], acc[mf][nf]);
                } else
#endif
#pragma unroll
                for (int nf = 0; nf < 3; ++nf)
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf)
                        if constexpr (!(H16_ABL & 8)) {
                            if constexpr (NPL == 2) {               // smallest partial product first
                                acc[mf][nf] = mma<true>(af[cur][1][mf], ring[rs][0][nf], acc[mf][nf]);
                                acc[mf][nf] = mma<true>(af[cur][0][mf], ring[rs][1][nf], acc[mf][nf]);
                                acc[mf][nf] = mma<true>(af[cur][0][mf], ring[rs][0][nf], acc[mf][nf]);
                            } else acc[mf][nf] = mma<F16>(af[(H16_ABL & 4) ? 0 : cur][0][mf], ring[rs][0][nf], acc[mf][nf]);
                        }
                if constexpr (!(H16_ABL & 1)) w_load(rs, S + RING < 18 ? cc : ccn, (S + RING) % 18);
                if constexpr (S >= H16_ST0 && S < H16_ST0 + NUT) {
                    if (!(H16_ABL & 2) && !(H16_ABL & 32)) a_store((c - c0 + 1) & 1, S - H16_ST0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 18>{});
        __syncthreads();
    }

    h16_epilogue(p, lds, acc, tid, lane, wm, wn, nb * 192, (long)tb, [&](int q, int pc) {
        const int wmm = pc >> 6, m2 = (pc >> 5) & 1, r = pc & 31;
        const int oy = y0 + 8 * wmm + 2 * (2 * q + m2) + (r >> 4), ox = x0 + (r & 15);
        return ((long)img * p.Hout + oy) * p.Wout + ox;
    });
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// k_conv_h2s: the fp16x2 3x3 kernel on an 8x16-pixel tile (128 pixels x 192 channels per workgroup, 96 accumulator registers per wave) so that TWO
// workgroups share a CU: the 16x16 kernel owns a CU alone (474 registers, 93 KB of LDS) and nothing overlaps its prologue, staging VALU and epilogue
// (no-MFMA ablation 106 of 461 us at 192 -> 192, 256 px, B = 4; the matrix pipe alone 336 - 350 us).  Same arithmetic, same weight image, same
// two-plane activation image; weight ring two k-steps deep; the epilogue exchange in two channel halves of 128 slots x 96 channels (50 KB).
// ---------------------------------------------------------------------------------------------------------------------------------------------
constexpr int H2S_PH = 10, H2S_NPIX = H2S_PH * H16_PW;               // patch of an 8x16 tile: 10 rows x 18 columns
constexpr int H2S_PLANE = H2S_PH * H16_LP * 64, H2S_STG = 2 * H2S_PLANE;   // bytes per plane / per stage (both planes)
constexpr int H2S_EP = 100;                                            // epilogue row pitch in floats (96 + 4)
constexpr int H2S_LDS = 2 * H2S_STG + 1024 > 128 * H2S_EP * 4 ? 2 * H2S_STG + 1024 : 128 * H2S_EP * 4;

template <class PixFn>
__device__ __forceinline__ void h2s_epilogue(const ConvK &p, char *lds, const f32x16 (&acc)[2][3], int tid, int lane, int wm, int wn, int n0, long tile, float rsx, PixFn pix) {
    float *ep = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                      // channel half h = the tiles of the waves with wn == h
        if (h) __syncthreads();
        if (wn == h) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int nf = 0; nf < 3; ++nf)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int r = (i >> 2) * 8 + (lane >> 5) * 4 + (i & 3);
                        ep[(wm * 64 + m2 * 32 + r) * H2S_EP + nf * 32 + (lane & 31)] = acc[m2][nf][i];
                    }
        }
        __syncthreads();
        // thread (pr0 = tid / 24 < 10, channel quad tid % 24 of this half) finishes pixel slots pr0, pr0 + 10, ... (<= 13 of the 128) in three batches:
        // the residual loads of a batch are in flight before its first store
        const int pr0 = tid / 24, cq = h * 96 + (tid - pr0 * 24) * 4, cl = cq - h * 96;
        f32x4 sm = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f}, sm2 = {0.f, 0.f, 0.f, 0.f}, sq2 = {0.f, 0.f, 0.f, 0.f};
        f32x4 bs = {0.f, 0.f, 0.f, 0.f}, ws = {1.f, 1.f, 1.f, 1.f};
        if (p.bias && pr0 < 10) bs = *reinterpret_cast<const f32x4 *>(p.bias + n0 + cq);
        if (p.wsc && pr0 < 10) ws = *reinterpret_cast<const f32x4 *>(p.wsc + n0 + cq);    // the inverse power-of-two scales of these four channels' weight planes ...
        ws = ws * rsx;                                                                    // ... and of the activation planes (both exact)
        auto finish = [&](auto kc0, auto nbc) {
            constexpr int K0 = decltype(kc0)::value, NB = decltype(nbc)::value;
            long mm[NB];
            f32x4 v[NB], rr[NB], r2[NB];
            bool on[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int pc = pr0 + 10 * (K0 + k);
                on[k] = pc < 128;
                mm[k] = pix(on[k] ? pc : 0);
                if (p.res && !p.partial && on[k]) rr[k] = *reinterpret_cast<const f32x4 *>(p.res + mm[k] * p.res_pitch + n0 + cq);
                if (p.out2 && !p.partial && on[k]) r2[k] = *reinterpret_cast<const f32x4 *>(p.res2 + mm[k] * p.res2_pitch + n0 + cq);
                v[k] = *reinterpret_cast<const f32x4 *>(ep + (on[k] ? pc : 0) * H2S_EP + cl) * ws;
            }
            if (p.partial) {
#pragma unroll
                for (int k = 0; k < NB; ++k) if (on[k]) *reinterpret_cast<f32x4 *>(p.partial + ((long)blockIdx.z * p.M + mm[k]) * p.Cout + n0 + cq) = v[k];
                return;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) if (on[k]) {
                v[k] += bs;
                if (p.res) v[k] += rr[k];
                sm += v[k]; sq += v[k] * v[k];
                *reinterpret_cast<f32x4 *>(p.out + mm[k] * p.out_pitch + n0 + cq) = v[k];
                if (p.out2) {
                    r2[k] += v[k];
                    sm2 += r2[k]; sq2 += r2[k] * r2[k];
                    *reinterpret_cast<f32x4 *>(p.out2 + mm[k] * p.out2_pitch + n0 + cq) = r2[k];
                }
            }
        };
        if (pr0 < 10) {
            finish(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
            finish(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{});
            finish(std::integral_constant<int, 9>{}, std::integral_constant<int, 4>{});
        }
        auto put_stats = [&](float *st, const f32x4 a_sm, const f32x4 a_sq) {   // (tile, half) = 128 pixels of one image x 96 channels; the ten pixel groups meet in LDS
            __syncthreads();
            if (pr0 < 10) {
                *reinterpret_cast<f32x4 *>(ep + ((pr0 * 24 + (cl >> 2)) * 2) * 4) = a_sm;
                *reinterpret_cast<f32x4 *>(ep + ((pr0 * 24 + (cl >> 2)) * 2 + 1) * 4) = a_sq;
            }
            __syncthreads();
            if (tid < 24) {
                f32x4 a = *reinterpret_cast<const f32x4 *>(ep + (tid * 2) * 4), b = *reinterpret_cast<const f32x4 *>(ep + (tid * 2 + 1) * 4);
#pragma unroll
                for (int g = 1; g < 10; ++g) {
                    a += *reinterpret_cast<const f32x4 *>(ep + ((g * 24 + tid) * 2) * 4);
                    b += *reinterpret_cast<const f32x4 *>(ep + ((g * 24 + tid) * 2 + 1) * 4);
                }
                const long img = (tile * 128) / ((long)p.Hout * p.Wout);
                const int c0_ = st == p.st1 ? p.st1_c0 : p.st2_c0, cg_ = st == p.st1 ? p.st1_cg : p.st2_cg;
                const int ch0 = c0_ + n0 + h * 96 + tid * 4;
                int g_run = ch0 / cg_;
                float s_run = a[0], q_run = b[0];
#pragma unroll
                for (int i = 1; i < 4; ++i) {
                    const int g_i = (ch0 + i) / cg_;
                    if (g_i != g_run) { stat_add(st, p.N, img, g_run, (long)p.Hout * p.Wout, s_run, q_run); g_run = g_i; s_run = 0.f; q_run = 0.f; }
                    s_run += a[i]; q_run += b[i];
                }
                stat_add(st, p.N, img, g_run, (long)p.Hout * p.Wout, s_run, q_run);
            }
        };
        if (p.st1 && !p.partial) put_stats(p.st1, sm, sq);
        if (p.st2 && !p.partial) put_stats(p.st2, sm2, sq2);
    }
}

// silu of a value the staging holds multiplied by the power of two sx: v = z sx  ->  silu(z) sx = v / (1 + 2^(kq v)), kq = -log2(e) / sx (sx = 1: silu_f of the GroupNorm passes)
__device__ __forceinline__ float h2s_silu(float v, float kq) { return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(kq * v)); }
__device__ __forceinline__ float wave_uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

__global__ __launch_bounds__(256, 2) void k_conv_h2s(const ConvK p) {
#if __HIP_DEVICE_COMPILE__
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NU = H2S_NPIX * 4, NUT = (NU + 255) / 256;      // staging units (pixel, 8-channel group): 720 -> 3 per thread
#ifndef H2S_RING
#define H2S_RING 2
#endif
#ifndef H2S_ABL   // timing ablations (wrong results): 1 no weight loads in the loop, 2 no patch loads / staging, 4 no A-fragment reads, 8 no MFMAs
#define H2S_ABL 0
#endif
#ifndef H2S_ST0
#define H2S_ST0 12
#endif
    constexpr int RING = H2S_RING;                                 // k-steps of weights in flight (divides 18)
    constexpr int ST0 = H2S_ST0;                                   // k-step behind which the next chunk's patch goes to LDS (3 units, one per step)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    const int tb = __builtin_amdgcn_readfirstlane(wi / p.n_nblocks), nb = wi - tb * p.n_nblocks;
    const int bw = p.Wout >> 4, bh = p.Hout >> 3;
    const int img = tb / (bw * bh), brem = tb - img * (bw * bh);
    const int y0 = (brem / bw) * 8, x0 = (brem - (brem / bw) * bw) * 16;
    const int nch = p.Cin >> 5;
    const unsigned pitch4 = (unsigned)p.in_pitch * (p.in16 ? 2u : 4u);
    const unsigned plane_b = (unsigned)((long)p.N * p.Hin * p.Win * p.in_pitch * 2);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.in, (short)0, (int)((long)p.N * p.Hin * p.Win * pitch4 * (p.in16 == 2 ? 2 : 1)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.w_bf3, (short)0, (int)((long)p.n_nblocks * nch * 18 * 6144 * 2), 0x00020000);

    // GroupNorm affine (+ SiLU) of the INPUT applied while staging (ConvK::cA / cB arrays, or ConvK::gn: coefficients formed here from the producers' group
    // totals): no pre-pass over the tensor.  The table sits behind the stages; padding pixels are zero AFTER the activation (the convolution pads the activated tensor).
    const bool aff = p.cA != nullptr || p.gn.gt != nullptr;
    float *sA = reinterpret_cast<float *>(lds + H2S_LDS), *sB = sA + p.Cin;
    float sx = 1.f, kq = -1.44269504088896341f;                    // power-of-two scale of the staged activations (hl_stats.h), -log2(e) / sx
    unsigned sv[NUT], sl[NUT];
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
        const int u = tid + 256 * j, pix = u >> 2, grp = u & 3;
        const int py = pix / H16_PW, px = pix - py * H16_PW;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        const bool ok = u < NU && y >= 0 && y < p.Hout && x >= 0 && x < p.Wout;
        const int ys = p.ups ? y >> 1 : y, xs = p.ups ? x >> 1 : x;
        sv[j] = ok ? (unsigned)((img * p.Hin + ys) * p.Win + xs) * pitch4 + grp * (p.in16 ? 16 : 32) : OOB;
        sl[j] = u < NU ? (unsigned)((py * H16_LP + px) * 64 + ((grp ^ ((px >> 2) & 3)) << 4)) : (unsigned)(2 * H2S_STG + (tid & 63) * 16);
    }
    u32x4 ar[NUT][2];
    auto a_load = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NUT; ++j) {
            const int so = chunk * (p.in16 ? 64 : 128);
            ar[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j], so, 0);
            ar[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j] == OOB ? OOB : sv[j] + (p.in16 ? plane_b : 16u), so, 0);
        }
    };
    auto a_store = [&](int stage, int j, int chunk) {
        u32x4 h0 = ar[j][0], h1 = ar[j][1];
        if (!p.in16) {
            f32x4 v0 = __builtin_bit_cast(f32x4, ar[j][0]), v1 = __builtin_bit_cast(f32x4, ar[j][1]);
            if (aff) {
                const int cb = chunk * 32 + (tid & 3) * 8;           // this unit's eight channels (the group is tid & 3 for every unit of the thread)
                v0 = v0 * *reinterpret_cast<const f32x4 *>(sA + cb) + *reinterpret_cast<const f32x4 *>(sB + cb);
                v1 = v1 * *reinterpret_cast<const f32x4 *>(sA + cb + 4) + *reinterpret_cast<const f32x4 *>(sB + cb + 4);
                if (p.act) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v0[i] = h2s_silu(v0[i], kq); v1[i] = h2s_silu(v1[i], kq); }
                }
                if (sv[j] == OOB) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
            } else { v0 = v0 * sx; v1 = v1 * sx; }
            const H2Pair q0 = split_h2(v0[0], v0[1]), q1 = split_h2(v0[2], v0[3]), q2 = split_h2(v1[0], v1[1]), q3 = split_h2(v1[2], v1[3]);
            h0 = u32x4{q0.p0, q1.p0, q2.p0, q3.p0}; h1 = u32x4{q0.p1, q1.p1, q2.p1, q3.p1};
        }
        char *dst = lds + (sl[j] >= 2u * H2S_STG ? 0 : stage * H2S_STG) + sl[j];
        *reinterpret_cast<u32x4 *>(dst) = h0;
        if (sl[j] < 2u * H2S_STG) *reinterpret_cast<u32x4 *>(dst + H2S_PLANE) = h1;
    };
    unsigned aoff[2][3];
    {
        const int r = lane & 31, g = lane >> 5;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int py = 4 * wm + 2 * mf + (r >> 4), px = (r & 15) + kx;
                aoff[mf][kx] = (unsigned)((py * H16_LP + px) * 64 + ((g ^ ((px >> 2) & 3)) << 4));
            }
    }
    const unsigned wv = (unsigned)lane * 16u;
    const int wbase = nb * nch * 18 * 6144 * 2 + wn * 3072;
    u32x4 ring[RING][2][3];
    const int c0 = (int)blockIdx.z * p.kt_per, c1 = min(nch, c0 + p.kt_per);
    auto w_load = [&](int slot_, int chunk, int s18) {
        const int so = wbase + (chunk * 18 + s18) * 6144 * 2;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) ring[slot_][pl][nf] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wv, so + pl * 6144 + nf * 1024, 0);
    };
    f32x16 acc[2][3];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;

    a_load(c0);
#pragma unroll
    for (int s = 0; s < RING; ++s) w_load(s, c0, s);
    if (aff) {   // (the first patch and weights are on their way)
        if (p.cA) {
            for (int c = tid; c < p.Cin; c += 256) { sA[c] = p.cA[(long)img * p.Cin + c]; sB[c] = p.cB[(long)img * p.Cin + c]; }
            __syncthreads();
        } else sx = wave_uniform(coef_to_lds<HL_ACT_SCALE != 0>(nullptr, nullptr, p.gn, p.N, img, sA, sB, sB + p.Cin, tid, 256));      // (ends with a barrier)
    } else if (HL_ACT_SCALE && p.xs_max && !p.in16) sx = wave_uniform(pow2_scale_for_bound(p.xs_max[img]));
    else if (HL_ACT_SCALE && p.xs_gt && !p.in16) sx = wave_uniform(act_scale_totals(p.xs_gt, p.N, img, p.xs_hw));
    const float rsx = 1.f / sx;
    kq *= rsx;
#pragma unroll
    for (int j = 0; j < NUT; ++j) a_store(0, j, c0);
    __syncthreads();

    u32x4 af[2][2];                                                // [plane][fragment mf] of the current k-step (read at its start: the partner wave of the SIMD covers the LDS latency)
    auto a_read = [&](const char *st, int tap, int k2, u32x4 (&dst)[2][2]) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) dst[pl][mf] = *reinterpret_cast<const u32x4 *>(st + pl * H2S_PLANE + (tap / 3) * (H16_LP * 64) + (aoff[mf][tap % 3] ^ (k2 ? 32u : 0u)));
    };
    for (int c = c0; c < c1; ++c) {
        const char *st = lds + ((c - c0) & 1) * H2S_STG;
        const int cc = c, ccn = cc + 1 < nch ? cc + 1 : 0;
        if (!(H2S_ABL & 2)) a_load(ccn);
        [&]<int... S>(std::integer_sequence<int, S...>) {
            ([&] {
                constexpr int rs = S % RING;
                if constexpr (!(H2S_ABL & 4) || S == 0) a_read(st, S >> 1, S & 1, af);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nf = 0; nf < 3; ++nf)
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) if constexpr (!(H2S_ABL & 8)) {                // smallest partial product first
                        acc[mf][nf] = mma<true>(af[1][mf], ring[rs][0][nf], acc[mf][nf]);
                        acc[mf][nf] = mma<true>(af[0][mf], ring[rs][1][nf], acc[mf][nf]);
                        acc[mf][nf] = mma<true>(af[0][mf], ring[rs][0][nf], acc[mf][nf]);
                    }
                if constexpr (!(H2S_ABL & 1)) w_load(rs, S + RING < 18 ? cc : ccn, (S + RING) % 18);
                if constexpr (S >= ST0 && S < ST0 + NUT && !(H2S_ABL & 2)) a_store((c - c0 + 1) & 1, S - ST0, ccn);
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 18>{});
        __syncthreads();
    }
    h2s_epilogue(p, lds, acc, tid, lane, wm, wn, nb * 192, (long)tb, rsx, [&](int pc) {
        const int wmm = pc >> 6, m2 = (pc >> 5) & 1, r = pc & 31;
        return ((long)img * p.Hout + y0 + 4 * wmm + 2 * m2 + (r >> 4)) * p.Wout + x0 + (r & 15);
    });
#endif
}

// k_conv1_h16: the 1x1 convolutions (skip / qkv / proj) in the same arithmetic.  A workgroup = 256 consecutive pixels x 192 output channels,
// the same 2x2 wave split, weight ring and epilogue; K in chunks of 96 input channels = six k-steps: the chunk's [256 px][96 ch] slab is fetched
// as fp32 one chunk ahead, rounded and written to LDS with a pixel pitch of 208 bytes (13 quarters: 13 is odd, so the 16 lanes a ds_read_b128
// is served in - 16 different pixels, one quarter - hit 16 different slots).  These layers are bound by HBM (a 384→192 layer at 256x256 moves
// 600 MB for 39 GFLOP), not by the matrix pipe.
constexpr int H1_PITCH = 208, H1_STAGE = 256 * H1_PITCH;

template <bool F16>
__global__ __launch_bounds__(256, 1) void k_conv1_h16(const ConvK p) {
#if __HIP_DEVICE_COMPILE__
    constexpr int NUT = 12;                                        // staging units per thread: 256 px x 12 groups of 8 channels / 256 threads
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    const int tb = __builtin_amdgcn_readfirstlane(wi / p.n_nblocks), nb = wi - tb * p.n_nblocks;
    const long m0 = (long)tb * 256;
    const int nch = p.Cin / 96;
    const unsigned pitch4 = (unsigned)p.in_pitch * (p.in16 ? 2u : 4u);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.in, (short)0, (int)((long)p.N * p.Hin * p.Win * pitch4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.w_bf3, (short)0, (int)((long)p.n_nblocks * nch * 6 * 6144), 0x00020000);

    unsigned sv[NUT], sl[NUT];
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
        const int u = tid + 256 * j, pix = u / 12, grp = u - pix * 12;
        sv[j] = (unsigned)(m0 + pix) * pitch4 + grp * (p.in16 ? 16 : 32);
        sl[j] = (unsigned)(pix * H1_PITCH + grp * 16);
    }
    u32x4 ar[NUT][2];
    auto a_load = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NUT; ++j) {
            ar[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j], chunk * (p.in16 ? 192 : 384), 0);
            if (!p.in16) ar[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j] + 16, chunk * 384, 0);
        }
    };
    auto a_store = [&](int stage, int j) {
        const f32x4 v0 = __builtin_bit_cast(f32x4, ar[j][0]), v1 = __builtin_bit_cast(f32x4, ar[j][1]);
        const u32x4 h = p.in16 ? ar[j][0]
                               : u32x4{pack2<F16>(v0[0], v0[1]), pack2<F16>(v0[2], v0[3]), pack2<F16>(v1[0], v1[1]), pack2<F16>(v1[2], v1[3])};
        *reinterpret_cast<u32x4 *>(lds + stage * H1_STAGE + sl[j]) = h;
    };
    unsigned aoff[4];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) aoff[mf] = (unsigned)((128 * wm + 32 * mf + (lane & 31)) * H1_PITCH + (lane >> 5) * 16);
    const unsigned wv = (unsigned)lane * 16u;
    const int wbase = nb * nch * 6 * 6144 + wn * 3072;
    u32x4 ring[6][3];
    auto w_load = [&](int slot_, int step) {
        const int so = wbase + min(step, nch * 6 - 1) * 6144;        // (past the end: the last step again, never used)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) ring[slot_][nf] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wv, so + nf * 1024, 0);
    };
    f32x16 acc[4][3];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;

    a_load(0);
#pragma unroll
    for (int s = 0; s < 6; ++s) w_load(s, s);
#pragma unroll
    for (int j = 0; j < NUT; ++j) a_store(0, j);
    __syncthreads();

    u32x4 af[2][4];
    auto a_read = [&](const char *st, int k2, u32x4 (&dst)[4]) {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) dst[mf] = *reinterpret_cast<const u32x4 *>(st + aoff[mf] + k2 * 32);
    };
    for (int c = 0; c < nch; ++c) {
        const char *st = lds + (c & 1) * H1_STAGE;
        a_load(c + 1 < nch ? c + 1 : c);                             // (last chunk: staged again, never read)
        a_read(st, 0, af[0]);
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... S>(std::integer_sequence<int, S...>) {
            ([&] {
                constexpr int cur = S & 1;
                if constexpr (S + 1 < 6) a_read(st, S + 1, af[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nf = 0; nf < 3; ++nf)
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf) acc[mf][nf] = mma<F16>(af[cur][mf], ring[S][nf], acc[mf][nf]);
                w_load(S, c * 6 + S + 6);
                a_store((c + 1) & 1, 2 * S);
                a_store((c + 1) & 1, 2 * S + 1);
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 6>{});
        __syncthreads();
    }
    if constexpr (H16_ABL & 128) { if (acc[0][0][0] == 12345.f) p.out[0] = acc[1][1][1] + acc[3][2][5]; return; }
    h16_epilogue(p, lds, acc, tid, lane, wm, wn, nb * 192, (long)tb, [&](int q, int pc) {
        return m0 + 128 * (pc >> 6) + 32 * (2 * q + ((pc >> 5) & 1)) + (pc & 31);
    });
#endif
}

// k_conv1_h2 (round 5): the 1x1 convolutions of the DEFAULT fp32 mode with fp16x2 products - every operand as two fp16 planes (activations
// h0 = the value with its low 13 mantissa bits cleared, h1 = the truncated residual: 2^-20; weights nearest-even planes at pack time: 2^-22),
// x w ~ h1 w0 + h0 w1 + h0 w0 accumulated in fp32 on v_mfma_f32_32x32x16_f16.  These layers (ResBlock skips, attention qkv / proj_out, the
// control tower's zero convolutions) are bound by HBM once they leave the fp32 matrix pipe (a 384->192 layer at 256x256 moves 600 MB for 39
// GFLOP): the three products cost nothing next to the memory time, and the fp32 kernel (k_conv_dma, 0.83 of the fp32 matrix peak) takes twice as
// long.  Error: a CPU emulation of exactly this arithmetic in ALL 1x1 layers of the production network moves its output by 3.1e-6 against the fp32
// oracle (scripts/unet_fp16x2_emulation.py; the fp32 kernels sit at 4.5e-6, the parity bound is 5e-5).  Same workgroup shape, wave split and
// epilogue as k_conv1_h16; K in chunks of 48 input channels = three k-steps (two planes of a chunk: 2 x 28 KB per stage), pixel pitch 112 bytes
// (7 quarters: odd), weights [channel block of 192][chunk of 48][k-step 3][plane 2][wn][fragment nf][lane][8], three k-steps in flight.
constexpr int H2_PITCH = 112, H2_PLANE = 256 * H2_PITCH, H2_STAGE = 2 * H2_PLANE;

__global__ __launch_bounds__(256, 1) void k_conv1_h2(const ConvK p) {
#if __HIP_DEVICE_COMPILE__
    constexpr int NUT = 6;                                         // staging units per thread: 256 px x 6 groups of 8 channels / 256 threads
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    const int tb = __builtin_amdgcn_readfirstlane(wi / p.n_nblocks), nb = wi - tb * p.n_nblocks;
    const long m0 = (long)tb * 256;
    const int nch = p.Cin / 48;
    const unsigned pitch4 = (unsigned)p.in_pitch * 4u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.in, (short)0, (int)((long)p.N * p.Hin * p.Win * pitch4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.w_bf3, (short)0, (int)((long)p.n_nblocks * nch * 3 * 12288), 0x00020000);

    unsigned sv[NUT], sl[NUT];
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
        const int u = tid + 256 * j, pix = u / 6, grp = u - pix * 6;
        sv[j] = (unsigned)(m0 + pix) * pitch4 + grp * 32;
        sl[j] = (unsigned)(pix * H2_PITCH + grp * 16);
    }
    u32x4 ar[NUT][2];
    auto a_load = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NUT; ++j) {
            ar[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j], chunk * 192, 0);
            ar[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j] + 16, chunk * 192, 0);
        }
    };
    auto a_store = [&](int stage, int j) {
        const f32x4 v0 = __builtin_bit_cast(f32x4, ar[j][0]), v1 = __builtin_bit_cast(f32x4, ar[j][1]);
        const H2Pair q0 = split_h2(v0[0], v0[1]), q1 = split_h2(v0[2], v0[3]), q2 = split_h2(v1[0], v1[1]), q3 = split_h2(v1[2], v1[3]);
        const u32x4 h0 = {q0.p0, q1.p0, q2.p0, q3.p0}, h1 = {q0.p1, q1.p1, q2.p1, q3.p1};
        *reinterpret_cast<u32x4 *>(lds + stage * H2_STAGE + sl[j]) = h0;
        *reinterpret_cast<u32x4 *>(lds + stage * H2_STAGE + H2_PLANE + sl[j]) = h1;
    };
    unsigned aoff[4];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) aoff[mf] = (unsigned)((128 * wm + 32 * mf + (lane & 31)) * H2_PITCH + (lane >> 5) * 16);
    const unsigned wv = (unsigned)lane * 16u;
    const int wbase = nb * nch * 3 * 12288 + wn * 3072;
    u32x4 ring[3][2][3];                                            // [k-step of the chunk][plane][fragment nf]
    auto w_load = [&](int slot_, int step) {
        const int so = wbase + min(step, nch * 3 - 1) * 12288;       // (past the end: the last step again, never used)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) ring[slot_][pl][nf] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wv, so + pl * 6144 + nf * 1024, 0);
    };
    f32x16 acc[4][3];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;

    a_load(0);
#pragma unroll
    for (int s = 0; s < 3; ++s) w_load(s, s);
#pragma unroll
    for (int j = 0; j < NUT; ++j) a_store(0, j);
    __syncthreads();

    u32x4 af[2][2][4];                                              // [buffer][plane][fragment mf]
    auto a_read = [&](const char *st, int k2, u32x4 (&dst)[2][4]) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) dst[pl][mf] = *reinterpret_cast<const u32x4 *>(st + pl * H2_PLANE + aoff[mf] + k2 * 32);
    };
    for (int c = 0; c < nch; ++c) {
        const char *st = lds + (c & 1) * H2_STAGE;
        if (!(H16_ABL & 2)) a_load(c + 1 < nch ? c + 1 : c);          // (last chunk: staged again, never read)
        a_read(st, 0, af[0]);
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... S>(std::integer_sequence<int, S...>) {
            ([&] {
                constexpr int cur = S & 1;
                if constexpr (S + 1 < 3 && !(H16_ABL & 4)) a_read(st, S + 1, af[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nf = 0; nf < 3; ++nf)
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf) if constexpr (!(H16_ABL & 8)) {                 // smallest partial product first
                        acc[mf][nf] = mma<true>(af[cur][1][mf], ring[S][0][nf], acc[mf][nf]);
                        acc[mf][nf] = mma<true>(af[cur][0][mf], ring[S][1][nf], acc[mf][nf]);
                        acc[mf][nf] = mma<true>(af[cur][0][mf], ring[S][0][nf], acc[mf][nf]);
                    }
                if constexpr (!(H16_ABL & 1)) w_load(S, c * 3 + S + 3);
                if constexpr (!(H16_ABL & 2)) { a_store((c + 1) & 1, 2 * S); a_store((c + 1) & 1, 2 * S + 1); }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 3>{});
        __syncthreads();
    }
    h16_epilogue(p, lds, acc, tid, lane, wm, wn, nb * 192, (long)tb, [&](int q, int pc) {
        return m0 + 128 * (pc >> 6) + 32 * (2 * q + ((pc >> 5) & 1)) + (pc & 31);
    });
#endif
}

// k_conv_h2d: the 3x3 / STRIDE-2 layers (Downsample, unet.py:100) in the same arithmetic: 8x16 OUTPUT pixels x 192 channels per workgroup, two workgroups per CU.  The 17x33-pixel
// input patch is staged de-interleaved by (row parity, column parity) into four 9x17 sub-images, so tap (ky, kx) of the 16 consecutive output pixels of a fragment row reads 16
// consecutive pixels of sub-image (ky & 1, kx & 1) at offset (ky >> 1, kx >> 1) - the access pattern of the stride-1 kernel.  Chunks of 16 input channels (32 bytes per pixel and
// plane, one k-step per tap), ONE stage of 46 KB: the next chunk's patch waits in registers while this one is multiplied (the partner workgroup of the CU covers the refill).
constexpr int H2D_SR = 9, H2D_LP = 20;                                      // rows / row pitch (pixels) of a sub-image
constexpr int H2D_SUB = H2D_SR * H2D_LP * 32, H2D_PLANE = 4 * H2D_SUB, H2D_STG = 2 * H2D_PLANE;   // bytes: sub-image, plane (four of them), stage (two planes)
constexpr int H2D_LDS = H2D_STG + 1024 > 128 * H2S_EP * 4 ? H2D_STG + 1024 : 128 * H2S_EP * 4;

__global__ __launch_bounds__(256, 2) void k_conv_h2d(const ConvK p) {
#if __HIP_DEVICE_COMPILE__
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NU = 17 * 33 * 2, NUT = (NU + 255) / 256;        // staging units (patch pixel, 8-channel group of the 16-channel chunk): 1 122 -> 5 per thread
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    const int tb = __builtin_amdgcn_readfirstlane(wi / p.n_nblocks), nb = wi - tb * p.n_nblocks;
    const int bw = p.Wout >> 4, bh = p.Hout >> 3;
    const int img = tb / (bw * bh), brem = tb - img * (bw * bh);
    const int y0 = (brem / bw) * 8, x0 = (brem - (brem / bw) * bw) * 16;
    const int nch = p.Cin >> 4;                                    // chunks of 16 input channels
    const unsigned pitch4 = (unsigned)p.in_pitch * 4u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, (short)0, (int)((long)p.N * p.Hin * p.Win * pitch4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)p.w_bf3, (short)0, (int)((long)p.n_nblocks * (p.Cin >> 5) * 18 * 6144 * 2), 0x00020000);
    unsigned sv[NUT], sl[NUT];
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
        const int u = tid + 256 * j, pix = u >> 1, grp = u & 1;
        const int py = pix / 33, px = pix - py * 33;
        const int y = 2 * y0 - 1 + py, x = 2 * x0 - 1 + px;
        const bool ok = u < NU && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
        sv[j] = ok ? (unsigned)((img * p.Hin + y) * p.Win + x) * pitch4 + grp * 32 : OOB;
        const int sub = (py & 1) * 2 + (px & 1), sy = py >> 1, sx = px >> 1;
        sl[j] = u < NU ? (unsigned)(sub * H2D_SUB + (sy * H2D_LP + sx) * 32 + ((grp ^ ((sx >> 3) & 1)) << 4)) : (unsigned)(H2D_STG + (tid & 63) * 16);   // (no unit: a dump slot)
    }
    u32x4 ar[NUT][2];
    auto a_load = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NUT; ++j) {
            ar[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j], chunk * 64, 0);
            ar[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j] == OOB ? OOB : sv[j] + 16u, chunk * 64, 0);
        }
    };
    // power-of-two scale of the raw input from its producers' totals (hl_stats.h)
    const float axs = !HL_ACT_SCALE ? 1.f : p.xs_max ? wave_uniform(pow2_scale_for_bound(p.xs_max[img])) : (p.xs_gt ? wave_uniform(act_scale_totals(p.xs_gt, p.N, img, p.xs_hw)) : 1.f), rsx = 1.f / axs;
    auto a_store = [&](int j) {
        const f32x4 v0 = __builtin_bit_cast(f32x4, ar[j][0]) * axs, v1 = __builtin_bit_cast(f32x4, ar[j][1]) * axs;
        const H2Pair q0 = split_h2(v0[0], v0[1]), q1 = split_h2(v0[2], v0[3]), q2 = split_h2(v1[0], v1[1]), q3 = split_h2(v1[2], v1[3]);
        *reinterpret_cast<u32x4 *>(lds + sl[j]) = u32x4{q0.p0, q1.p0, q2.p0, q3.p0};
        if (sl[j] < (unsigned)H2D_STG) *reinterpret_cast<u32x4 *>(lds + H2D_PLANE + sl[j]) = u32x4{q0.p1, q1.p1, q2.p1, q3.p1};
    };
    // A fragments: row r of fragment mf = output pixel (4 wm + 2 mf + (r >> 4), r & 15); column offset dx = kx >> 1 in its sub-image
    unsigned aoff[2][2];
    {
        const int r = lane & 31, g = lane >> 5;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int sy = 4 * wm + 2 * mf + (r >> 4), sx = (r & 15) + dx;
                aoff[mf][dx] = (unsigned)((sy * H2D_LP + sx) * 32 + ((g ^ ((sx >> 3) & 1)) << 4));
            }
    }
    const unsigned wv = (unsigned)lane * 16u;
    const int wbase = nb * (p.Cin >> 5) * 18 * 6144 * 2 + wn * 3072;
    u32x4 ring[3][2][3];                                           // three taps of weights in flight (a chunk has nine)
    auto w_load = [&](int slot_, int c16, int tap) {               // k-step (tap, half c16 & 1) of the 32-channel chunk c16 >> 1 of the packed image
        const int so = wbase + ((c16 >> 1) * 18 + tap * 2 + (c16 & 1)) * 6144 * 2;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) ring[slot_][pl][nf] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wv, so + pl * 6144 + nf * 1024, 0);
    };
    f32x16 acc[2][3];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;
    a_load(0);
#pragma unroll
    for (int s = 0; s < 3; ++s) w_load(s, 0, s);
    u32x4 af[2][2];
    for (int c = 0; c < nch; ++c) {
        if (c) __syncthreads();                                      // everybody is done with the previous chunk's patch
#pragma unroll
        for (int j = 0; j < NUT; ++j) a_store(j);
        __syncthreads();
        const int cn = c + 1 < nch ? c + 1 : c;                      // (past the end: a valid chunk, fetched and never written)
        a_load(cn);
        [&]<int... T>(std::integer_sequence<int, T...>) {
            ([&] {
                constexpr int ky = T / 3, kx = T % 3, rs = T % 3;
                const char *sb = lds + ((ky & 1) * 2 + (kx & 1)) * H2D_SUB + (ky >> 1) * (H2D_LP * 32);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) af[pl][mf] = *reinterpret_cast<const u32x4 *>(sb + pl * H2D_PLANE + aoff[mf][kx >> 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nf = 0; nf < 3; ++nf)
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) {                // smallest partial product first
                        acc[mf][nf] = mma<true>(af[1][mf], ring[rs][0][nf], acc[mf][nf]);
                        acc[mf][nf] = mma<true>(af[0][mf], ring[rs][1][nf], acc[mf][nf]);
                        acc[mf][nf] = mma<true>(af[0][mf], ring[rs][0][nf], acc[mf][nf]);
                    }
                w_load(rs, T + 3 < 9 ? c : cn, (T + 3) % 9);
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 9>{});
    }
    __syncthreads();
    h2s_epilogue(p, lds, acc, tid, lane, wm, wn, nb * 192, (long)tb, rsx, [&](int pc) {
        const int wmm = pc >> 6, m2 = (pc >> 5) & 1, r = pc & 31;
        return ((long)img * p.Hout + y0 + 4 * wmm + 2 * m2 + (r >> 4)) * p.Wout + x0 + (r & 15);
    });
#endif
}

// k_conv1_h2s: k_conv1_h2 on 128 pixels x 192 channels per workgroup (96 accumulator registers per wave, 56 KB of LDS): TWO workgroups share a CU, so the
// prologue (the first slab's HBM round trip) and the epilogue of one overlap the main loop of the other - with one 256-pixel workgroup per CU they were ~14 of
// ~46 us per workgroup (timing ablations, profiles/r05_unet_fill_experiments.md section 7).  Same arithmetic, same weight image, h2s_epilogue.
constexpr int H2S1_PLANE = 128 * H2_PITCH, H2S1_STAGE = 2 * H2S1_PLANE;          // 14 KB per plane, 28 KB per stage
constexpr int H2S1_LDS = 2 * H2S1_STAGE > 128 * H2S_EP * 4 ? 2 * H2S1_STAGE : 128 * H2S_EP * 4;

__global__ __launch_bounds__(256, 2) void k_conv1_h2s(const ConvK p) {
#if __HIP_DEVICE_COMPILE__
    constexpr int NUT = 3;                                         // staging units per thread: 128 px x 6 groups of 8 channels / 256 threads
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    const int tb = __builtin_amdgcn_readfirstlane(wi / p.n_nblocks), nb = wi - tb * p.n_nblocks;
    const long m0 = (long)tb * 128;
    const int nch = p.Cin / 48;
    const int c0 = (int)blockIdx.z * p.kt_per, c1 = min(nch, c0 + p.kt_per);      // split-K: blockIdx.z owns the chunks [c0, c1), raw sums go to p.partial
    const unsigned pitch4 = (unsigned)p.in_pitch * 4u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.in, (short)0, (int)((long)p.N * p.Hin * p.Win * pitch4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.w_bf3, (short)0, (int)((long)p.n_nblocks * nch * 3 * 12288), 0x00020000);
    // GroupNorm affine (+ SiLU) of the input applied while staging, as in k_conv_h2s (ConvK::cA / cB arrays or ConvK::gn); a tile = 128 pixels of ONE image
    const bool aff = p.cA != nullptr || p.gn.gt != nullptr;
    float *sA = reinterpret_cast<float *>(lds + H2S1_LDS), *sB = sA + p.Cin;
    const int img = (int)(m0 / ((long)p.Hout * p.Wout));
    float sx = 1.f, kq = -1.44269504088896341f;                    // power-of-two scale of the staged activations (hl_stats.h), -log2(e) / sx
    unsigned sv[NUT], sl[NUT];
    int sg[NUT];
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
        const int u = tid + 256 * j, pix = u / 6, grp = u - pix * 6;
        sg[j] = grp * 8;
        sv[j] = (unsigned)(m0 + pix) * pitch4 + grp * 32;
        sl[j] = (unsigned)(pix * H2_PITCH + grp * 16);
    }
    u32x4 ar[NUT][2];
    auto a_load = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NUT; ++j) {
            ar[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j], chunk * 192, 0);
            ar[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sv[j] + 16, chunk * 192, 0);
        }
    };
    auto a_store = [&](int stage, int j, int chunk) {
        f32x4 v0 = __builtin_bit_cast(f32x4, ar[j][0]), v1 = __builtin_bit_cast(f32x4, ar[j][1]);
        if (aff) {
            const int cb = chunk * 48 + sg[j];
            v0 = v0 * *reinterpret_cast<const f32x4 *>(sA + cb) + *reinterpret_cast<const f32x4 *>(sB + cb);
            v1 = v1 * *reinterpret_cast<const f32x4 *>(sA + cb + 4) + *reinterpret_cast<const f32x4 *>(sB + cb + 4);
            if (p.act) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] = h2s_silu(v0[i], kq); v1[i] = h2s_silu(v1[i], kq); }
            }
        }
        else { v0 = v0 * sx; v1 = v1 * sx; }
        const H2Pair q0 = split_h2(v0[0], v0[1]), q1 = split_h2(v0[2], v0[3]), q2 = split_h2(v1[0], v1[1]), q3 = split_h2(v1[2], v1[3]);
        const u32x4 h0 = {q0.p0, q1.p0, q2.p0, q3.p0}, h1 = {q0.p1, q1.p1, q2.p1, q3.p1};
        *reinterpret_cast<u32x4 *>(lds + stage * H2S1_STAGE + sl[j]) = h0;
        *reinterpret_cast<u32x4 *>(lds + stage * H2S1_STAGE + H2S1_PLANE + sl[j]) = h1;
    };
    unsigned aoff[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) aoff[mf] = (unsigned)((64 * wm + 32 * mf + (lane & 31)) * H2_PITCH + (lane >> 5) * 16);
    const unsigned wv = (unsigned)lane * 16u;
    const int wbase = nb * nch * 3 * 12288 + wn * 3072;
    u32x4 ring[3][2][3];                                            // [k-step of the chunk][plane][fragment nf]
    auto w_load = [&](int slot_, int step) {
        const int so = wbase + min(step, nch * 3 - 1) * 12288;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) ring[slot_][pl][nf] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wv, so + pl * 6144 + nf * 1024, 0);
    };
    f32x16 acc[2][3];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;
    a_load(c0);
#pragma unroll
    for (int s = 0; s < 3; ++s) w_load(s, c0 * 3 + s);
    if (aff) {   // (the first slab and weights are on their way)
        if (p.cA) {
            for (int c = tid; c < p.Cin; c += 256) { sA[c] = p.cA[(long)img * p.Cin + c]; sB[c] = p.cB[(long)img * p.Cin + c]; }
            __syncthreads();
        } else sx = wave_uniform(coef_to_lds<HL_ACT_SCALE != 0>(nullptr, nullptr, p.gn, p.N, img, sA, sB, sB + p.Cin, tid, 256));      // (ends with a barrier)
    } else if (HL_ACT_SCALE && p.xs_max && !p.in16) sx = wave_uniform(pow2_scale_for_bound(p.xs_max[img]));
    else if (HL_ACT_SCALE && p.xs_gt && !p.in16) sx = wave_uniform(act_scale_totals(p.xs_gt, p.N, img, p.xs_hw));
    const float rsx = 1.f / sx;
    kq *= rsx;
#pragma unroll
    for (int j = 0; j < NUT; ++j) a_store(0, j, c0);
    __syncthreads();
    u32x4 af[2][2];                                                 // [plane][fragment mf] of the current k-step (the partner wave of the SIMD covers the LDS latency)
    for (int c = c0; c < c1; ++c) {
        const char *st = lds + ((c - c0) & 1) * H2S1_STAGE;
        const int cn = c + 1 < c1 ? c + 1 : c;                      // (last chunk: staged again, never read)
        a_load(cn);
        [&]<int... S>(std::integer_sequence<int, S...>) {
            ([&] {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) af[pl][mf] = *reinterpret_cast<const u32x4 *>(st + pl * H2S1_PLANE + aoff[mf] + S * 32);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nf = 0; nf < 3; ++nf)
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) {                 // smallest partial product first
                        acc[mf][nf] = mma<true>(af[1][mf], ring[S][0][nf], acc[mf][nf]);
                        acc[mf][nf] = mma<true>(af[0][mf], ring[S][1][nf], acc[mf][nf]);
                        acc[mf][nf] = mma<true>(af[0][mf], ring[S][0][nf], acc[mf][nf]);
                    }
                w_load(S, c * 3 + S + 3);
                a_store((c - c0 + 1) & 1, S, cn);
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 3>{});
        __syncthreads();
    }
    h2s_epilogue(p, lds, acc, tid, lane, wm, wn, nb * 192, (long)tb, rsx, [&](int pc) { return m0 + pc; });
#endif
}

// Scale of the fp16x2 weight planes (round 6).  fp16 has a 5-bit exponent: unscaled, the low plane w1 = w - fp16(w) of a weight below 2^-3 falls under
// fp16's smallest normal (2^-14) and is rounded on the 2^-24 subnormal grid - at |w| ~ 5e-3 (the production net's own 1536 -> 768 layers at initialisation)
// the pair keeps 2^-18 instead of 2^-22, at 1e-4 (zero_module convolutions early in training, unet.py:149,237) TF32's 2^-11.  So every OUTPUT CHANNEL's
// weights are multiplied, before they are split, by the power of two that puts the channel's largest |w| into [2^13, 2^14): both planes are then normal
// for every weight down to 2^-16 of the channel's largest, the pair keeps 2^-22 of it whatever the magnitude, and the scaling itself is exact.  The
// epilogues multiply the accumulators by the inverse (a power of two: exact) before the bias; the inverses live behind the planes of the layer
// (conv_packed_h2_bytes counts them, h2_wscale finds them).  A channel of zeros, or one whose largest weight is not finite, keeps scale 1.
__global__ void k_wscale_h2(const float *__restrict__ w, int Cout, int Cin, int taps, int tf, float *__restrict__ inv) {
    __shared__ float red[256];
    const int o = blockIdx.x;
    float m = 0.f;
    for (int i = threadIdx.x; i < Cin * taps; i += 256) {
        const int cin = i / taps, tap = i - cin * taps;
        const float v = tf ? w[((long)cin * Cout + o) * taps + tap] : w[((long)o * Cin + cin) * taps + tap];
        m = fmaxf(m, fabsf(v));          // (fmaxf drops NaNs: a NaN weight still reaches the planes and the output through the pack below)
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        m = red[0];
        int e = (int)((__float_as_uint(m) >> 23) & 0xff) - 127;        // floor(log2 m) for a normal m
        float r = 1.f;
        if (m > 0.f && e < 128) {                                      // (e == 128: inf)
            e = e < -100 ? -100 : e;                                   // (denormal weights: 2^113 at most)
            r = __uint_as_float((unsigned)(e - 13 + 127) << 23);       // 2^(e - 13): m / r is in [2^13, 2^14)
        }
        inv[o] = r;
    }
}
__device__ __forceinline__ unsigned short h2_plane(float v, float inv, int pl) {   // plane pl of v / inv, nearest even at both levels
    v = v * (1.f / inv);                       // (1 / a power of two: exact)
    _Float16 h = (_Float16)v;
    if (pl) h = (_Float16)(v - (float)h);
    return __builtin_bit_cast(unsigned short, h);
}

// fp16x2 weights of a 3x3 layer: [channel block of 192][chunk of 32 inputs][tap][k-half][plane 2][wn][fragment nf][lane][8], then Cout floats = the inverse scales
__global__ void k_pack_conv_h2(const float *__restrict__ w, int Cout, int Cin, int Cin_pad, unsigned short *__restrict__ dst, int tf, const float *__restrict__ inv) {
    const int nch = Cin_pad >> 5;
    const long n = (long)(Cout / 192) * nch * 18 * 2 * 3072;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
        long t = i >> 9;
        const int nf = (int)(t % 3); t /= 3;
        const int wn = (int)(t & 1); t >>= 1;
        const int pl = (int)(t & 1); t >>= 1;
        const int k2 = (int)(t & 1); t >>= 1;
        const int tap = (int)(t % 9); t /= 9;
        const int chunk = (int)(t % nch);
        const int nb = (int)(t / nch);
        const int o = nb * 192 + wn * 96 + nf * 32 + (l & 31), cin = chunk * 32 + k2 * 16 + (l >> 5) * 8 + j;
        float v = 0.f;
        if (cin < Cin) v = tf ? w[((long)cin * Cout + o) * 9 + (8 - tap)] : w[((long)o * Cin + cin) * 9 + tap];
        dst[i] = h2_plane(v, inv[o], pl);
    }
}

// fp16x2 weights of a 1x1 layer: [channel block of 192][chunk of 48 inputs][k-step 3][plane 2][wn][fragment nf][lane][8], then Cout floats = the inverse scales
__global__ void k_pack_conv1_h2(const float *__restrict__ w, int Cout, int Cin, int Cin_pad, unsigned short *__restrict__ dst, int tf, const float *__restrict__ inv) {
    const int nch = Cin_pad / 48;
    const long n = (long)(Cout / 192) * nch * 3 * 2 * 3072;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
        long t = i >> 9;
        const int nf = (int)(t % 3); t /= 3;
        const int wn = (int)(t & 1); t >>= 1;
        const int pl = (int)(t & 1); t >>= 1;
        const int k2 = (int)(t % 3); t /= 3;
        const int chunk = (int)(t % nch);
        const int nb = (int)(t / nch);
        const int o = nb * 192 + wn * 96 + nf * 32 + (l & 31), cin = chunk * 48 + k2 * 16 + (l >> 5) * 8 + j;
        float v = 0.f;
        if (cin < Cin) v = tf ? w[(long)cin * Cout + o] : w[(long)o * Cin + cin];
        dst[i] = h2_plane(v, inv[o], pl);
    }
}

// weights -> 16-bit values in fragment order: [channel block of 192][chunk of 32 inputs][tap][k-half][wn][fragment nf][lane][8]
// lane l of fragment (wn, nf) holds output channel 192 nb + 96 wn + 32 nf + (l & 31), inputs 32 chunk + 16 k-half + 8 (l >> 5) + 0..7
__global__ void k_pack_conv_h16(const float *__restrict__ w, int Cout, int Cin, int Cin_pad, unsigned short *__restrict__ dst, int f16, int tf) {
    const int nch = Cin_pad >> 5;
    const long n = (long)(Cout / 192) * nch * 18 * 3072;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
        long t = i >> 9;
        const int nf = (int)(t % 3); t /= 3;
        const int wn = (int)(t & 1); t >>= 1;
        const int k2 = (int)(t & 1); t >>= 1;
        const int tap = (int)(t % 9); t /= 9;
        const int chunk = (int)(t % nch);
        const int nb = (int)(t / nch);
        const int o = nb * 192 + wn * 96 + nf * 32 + (l & 31), cin = chunk * 32 + k2 * 16 + (l >> 5) * 8 + j;
        float v = 0.f;
        if (cin < Cin) v = tf ? w[((long)cin * Cout + o) * 9 + (8 - tap)] : w[((long)o * Cin + cin) * 9 + tap];
        unsigned short h;
        if (f16) {
            const _Float16 hh = (_Float16)v;
            h = __builtin_bit_cast(unsigned short, hh);
        } else {
            const unsigned u = __float_as_uint(v);
            h = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        }
        dst[i] = h;
    }
}

// 1x1 layers: [channel block of 192][chunk of 96 inputs][k-step 6][wn][fragment nf][lane][8]
__global__ void k_pack_conv1_h16(const float *__restrict__ w, int Cout, int Cin, int Cin_pad, unsigned short *__restrict__ dst, int f16, int tf) {
    const int nch = Cin_pad / 96;
    const long n = (long)(Cout / 192) * nch * 6 * 3072;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
        long t = i >> 9;
        const int nf = (int)(t % 3); t /= 3;
        const int wn = (int)(t & 1); t >>= 1;
        const int k2 = (int)(t % 6); t /= 6;
        const int chunk = (int)(t % nch);
        const int nb = (int)(t / nch);
        const int o = nb * 192 + wn * 96 + nf * 32 + (l & 31), cin = chunk * 96 + k2 * 16 + (l >> 5) * 8 + j;
        float v = 0.f;
        if (cin < Cin) v = tf ? w[(long)cin * Cout + o] : w[(long)o * Cin + cin];
        unsigned short h;
        if (f16) {
            const _Float16 hh = (_Float16)v;
            h = __builtin_bit_cast(unsigned short, hh);
        } else {
            const unsigned u = __float_as_uint(v);
            h = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        }
        dst[i] = h;
    }
}

}  // namespace

bool conv_h16_applies(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int ups) {
    // 3x3: the nearest-x2 upsample in front of the convolution is a source-address shift of the patch loads; 1x1: 256 consecutive pixels per tile
    if (ks == 3) return stride == 1 && Hout % 16 == 0 && Wout % 16 == 0 && Cin % 32 == 0 && Cout % 192 == 0 && (long)Cout * Cin * 18 < (1L << 31);
    return ks == 1 && stride == 1 && !ups && ((long)Hout * Wout) % 256 == 0 && Cin % 96 == 0 && Cout % 192 == 0 && (long)Cout * Cin * 2 < (1L << 31);
}

size_t conv_packed_h16_bytes(int Cout, int Cin_pad, int ks) {
    if (ks == 3) return (Cout % 192 == 0 && Cin_pad % 32 == 0) ? (size_t)Cout * Cin_pad * 9 * 2 : 0;
    return (ks == 1 && Cout % 192 == 0 && Cin_pad % 96 == 0) ? (size_t)Cout * Cin_pad * 2 : 0;
}

int conv_pack_weights_h16(const float *w, int Cout, int Cin, int Cin_pad, int ks, void *packed, int f16, hipStream_t st, int tf) {
    HL_REQUIRE(w && packed && conv_packed_h16_bytes(Cout, Cin_pad, ks) && Cin <= Cin_pad, "conv_pack_weights_h16: bad argument");
    if (ks == 3) hipLaunchKernelGGL(k_pack_conv_h16, dim3(1024), dim3(256), 0, st, w, Cout, Cin, Cin_pad, static_cast<unsigned short *>(packed), f16, tf);
    else hipLaunchKernelGGL(k_pack_conv1_h16, dim3(512), dim3(256), 0, st, w, Cout, Cin, Cin_pad, static_cast<unsigned short *>(packed), f16, tf);
    return check_launch("k_pack_conv_h16");
}

bool conv1_h2_applies(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int ups) {
    return ks == 1 && stride == 1 && !ups && ((long)Hout * Wout) % 256 == 0 && Cin % 48 == 0 && Cout % 192 == 0 && (long)Cout * Cin * 4 < (1L << 31);
}
static size_t h2_plane_bytes(int Cout, int Cin_pad, int ks) {
    if (ks == 3) return (Cout % 192 == 0 && Cin_pad % 32 == 0) ? (size_t)Cout * Cin_pad * 9 * 4 : 0;
    return (ks == 1 && Cout % 192 == 0 && Cin_pad % 48 == 0) ? (size_t)Cout * Cin_pad * 4 : 0;
}
size_t conv_packed_h2_bytes(int Cout, int Cin_pad, int ks) {   // the two planes + the per-channel inverse scales (k_wscale_h2)
    const size_t b = h2_plane_bytes(Cout, Cin_pad, ks);
    return b ? b + (size_t)Cout * sizeof(float) : 0;
}
const float *conv_h2_wscale(const void *packed, int Cout, int Cin_pad, int ks) {
    return reinterpret_cast<const float *>(static_cast<const char *>(packed) + h2_plane_bytes(Cout, Cin_pad, ks));
}
int conv_pack_weights_h2(const float *w, int Cout, int Cin, int Cin_pad, int ks, void *packed, hipStream_t st, int tf) {
    HL_REQUIRE(w && packed && conv_packed_h2_bytes(Cout, Cin_pad, ks) && Cin <= Cin_pad, "conv_pack_weights_h2: bad argument");
    float *inv = const_cast<float *>(conv_h2_wscale(packed, Cout, Cin_pad, ks));
    hipLaunchKernelGGL(k_wscale_h2, dim3((unsigned)Cout), dim3(256), 0, st, w, Cout, Cin, ks * ks, tf, inv);
    if (ks == 3) hipLaunchKernelGGL(k_pack_conv_h2, dim3(1024), dim3(256), 0, st, w, Cout, Cin, Cin_pad, static_cast<unsigned short *>(packed), tf, inv);
    else hipLaunchKernelGGL(k_pack_conv1_h2, dim3(512), dim3(256), 0, st, w, Cout, Cin, Cin_pad, static_cast<unsigned short *>(packed), tf, inv);
    return check_launch("k_pack_conv_h2");
}
// the 3x3 / stride-1 layers with fp16x2 products (k_conv_h16<true, 2>)
static size_t conv3_h2_lds_bytes() { return (size_t)2 * H1_STAGE; }   // (two stages of two planes + the dump slot = 93 KB; the epilogue exchange 98 KB)
static_assert(4 * H16_STAGE + 1024 <= 2 * H1_STAGE, "k_conv_h2 stages");
int conv3_h2_launch(const ConvK &p, hipStream_t st, int splits) {
    HL_REQUIRE(p.w_bf3 && p.ks == 3 && conv_h16_applies(p.Hout, p.Wout, p.Cin, p.Cout, p.ks, p.stride, p.ups) && (splits == 1 || p.partial), "k_conv_h2: bad layer");
    const size_t sh = conv3_h2_lds_bytes();
    static const bool attr_ok = hipFuncSetAttribute((const void *)k_conv_h16<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv3_h2_lds_bytes()) == hipSuccess;
    HL_REQUIRE(attr_ok, "k_conv_h2: cannot raise the dynamic LDS limit to %zu bytes", sh);
    hipLaunchKernelGGL((k_conv_h16<true, 2>), dim3((unsigned)(p.n_mtiles * p.n_nblocks), 1, (unsigned)splits), dim3(256), sh, st, p);
    return check_launch("k_conv_h2");
}

// the same layers on the 8x16-pixel tile (two workgroups per CU): p.n_mtiles = pixels / 128
int conv3_h2s_launch(const ConvK &p, hipStream_t st, int splits) {
    HL_REQUIRE(p.w_bf3 && p.ks == 3 && conv_h16_applies(p.Hout, p.Wout, p.Cin, p.Cout, p.ks, p.stride, p.ups) && (splits == 1 || p.partial), "k_conv_h2s: bad layer");
    constexpr int LDS_MAX = H2S_LDS + (2 * 4096 + COEF_SCR_FLOATS) * 4;         // + the coefficient table of a fused GroupNorm (<= 4096 input channels)
    static const bool attr_ok = hipFuncSetAttribute((const void *)k_conv_h2s, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) == hipSuccess;
    HL_REQUIRE(attr_ok, "k_conv_h2s: cannot raise the dynamic LDS limit to %d bytes", LDS_MAX);
    const bool aff = p.cA != nullptr || p.gn.gt != nullptr;
    HL_REQUIRE(!aff || (p.Cin <= 4096 && !p.in16 && !p.ups), "k_conv_h2s: fused GroupNorm needs a raw fp32 input of <= 4096 channels");
    const size_t lds_bytes = (size_t)H2S_LDS + (aff ? (size_t)(2 * p.Cin + COEF_SCR_FLOATS) * 4 : 0);
    hipLaunchKernelGGL(k_conv_h2s, dim3((unsigned)(p.n_mtiles * p.n_nblocks), 1, (unsigned)splits), dim3(256), lds_bytes, st, p);
    return check_launch("k_conv_h2s");
}

// the 3x3 / stride-2 layers: p.n_mtiles = output pixels / 128
bool conv3_h2d_applies(int Hout, int Wout, int Cin, int Cout) { return Hout % 8 == 0 && Wout % 16 == 0 && Cin % 32 == 0 && Cout % 192 == 0 && (long)Cout * Cin * 36 < (1L << 31); }
int conv3_h2d_launch(const ConvK &p, hipStream_t st) {
    HL_REQUIRE(p.w_bf3 && p.ks == 3 && p.stride == 2 && !p.ups && !p.in16 && !p.partial && conv3_h2d_applies(p.Hout, p.Wout, p.Cin, p.Cout) && p.cA == nullptr && p.gn.gt == nullptr,
               "k_conv_h2d: bad layer");
    static const bool attr_ok = hipFuncSetAttribute((const void *)k_conv_h2d, hipFuncAttributeMaxDynamicSharedMemorySize, H2D_LDS) == hipSuccess;
    HL_REQUIRE(attr_ok, "k_conv_h2d: cannot raise the dynamic LDS limit to %d bytes", H2D_LDS);
    hipLaunchKernelGGL(k_conv_h2d, dim3((unsigned)(p.n_mtiles * p.n_nblocks)), dim3(256), (size_t)H2D_LDS, st, p);
    return check_launch("k_conv_h2d");
}

// the 1x1 layers on 128-pixel tiles (two workgroups per CU): p.n_mtiles = pixels / 128
int conv1_h2s_launch(const ConvK &p, hipStream_t st, int splits) {
    HL_REQUIRE(p.w_bf3 && conv1_h2_applies(p.Hout, p.Wout, p.Cin, p.Cout, p.ks, p.stride, p.ups) && (splits == 1 || p.partial) && !p.in16 && p.kt_per >= 1, "k_conv1_h2s: bad layer");
    constexpr int LDS_MAX = H2S1_LDS + (2 * 4096 + COEF_SCR_FLOATS) * 4;        // + the coefficient table of a fused GroupNorm (<= 4096 input channels)
    static const bool attr_ok = hipFuncSetAttribute((const void *)k_conv1_h2s, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) == hipSuccess;
    HL_REQUIRE(attr_ok, "k_conv1_h2s: cannot raise the dynamic LDS limit to %d bytes", LDS_MAX);
    const bool aff = p.cA != nullptr || p.gn.gt != nullptr;
    HL_REQUIRE(!aff || (p.Cin <= 4096 && ((long)p.Hout * p.Wout) % 128 == 0), "k_conv1_h2s: fused GroupNorm needs <= 4096 input channels and whole tiles per image");
    const size_t lds_bytes = (size_t)H2S1_LDS + (aff ? (size_t)(2 * p.Cin + COEF_SCR_FLOATS) * 4 : 0);
    hipLaunchKernelGGL(k_conv1_h2s, dim3((unsigned)(p.n_mtiles * p.n_nblocks), 1, (unsigned)splits), dim3(256), lds_bytes, st, p);
    return check_launch("k_conv1_h2s");
}

int conv1_h2_launch(const ConvK &p, hipStream_t st) {
    HL_REQUIRE(p.w_bf3 && conv1_h2_applies(p.Hout, p.Wout, p.Cin, p.Cout, p.ks, p.stride, p.ups) && !p.in16 && !p.partial, "k_conv1_h2: bad layer");
    static const bool attr_ok = hipFuncSetAttribute((const void *)k_conv1_h2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * H2_STAGE)) == hipSuccess;
    HL_REQUIRE(attr_ok, "k_conv1_h2: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(k_conv1_h2, dim3((unsigned)(p.n_mtiles * p.n_nblocks)), dim3(256), (size_t)2 * H2_STAGE, st, p);   // 112 KB (the epilogue exchange needs 98 KB)
    return check_launch("k_conv1_h2");
}

size_t conv_h16_lds_bytes() { return (size_t)2 * H1_STAGE; }   // 104 KB: the two 1x1 stages (the epilogue exchange needs 98 KB, the 3x3 patch stages 45 KB)

int conv_h16_launch(const ConvK &p, int f16, hipStream_t st, int splits) {
    HL_REQUIRE(p.w_bf3 && conv_h16_applies(p.Hout, p.Wout, p.Cin, p.Cout, p.ks, p.stride, p.ups), "k_conv_h16: bad layer");
    HL_REQUIRE(splits == 1 || (p.ks == 3 && p.partial), "k_conv_h16: split-K is the 3x3 kernel's");
    const dim3 grid((unsigned)(p.n_mtiles * p.n_nblocks), 1, (unsigned)splits);
    const size_t sh = conv_h16_lds_bytes();
    static const bool attr_ok = [] {
        const int b = (int)conv_h16_lds_bytes();
        return hipFuncSetAttribute((const void *)k_conv_h16<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, b) == hipSuccess &&
               hipFuncSetAttribute((const void *)k_conv_h16<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, b) == hipSuccess &&
               hipFuncSetAttribute((const void *)k_conv1_h16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, b) == hipSuccess &&
               hipFuncSetAttribute((const void *)k_conv1_h16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, b) == hipSuccess;
    }();
    HL_REQUIRE(attr_ok, "k_conv_h16: cannot raise the dynamic LDS limit to %zu bytes", sh);
    if (p.ks == 1) {
        if (f16) hipLaunchKernelGGL((k_conv1_h16<true>), grid, dim3(256), sh, st, p);
        else hipLaunchKernelGGL((k_conv1_h16<false>), grid, dim3(256), sh, st, p);
        return check_launch("k_conv1_h16");
    }
    if (f16) hipLaunchKernelGGL((k_conv_h16<true, 1>), grid, dim3(256), sh, st, p);
    else hipLaunchKernelGGL((k_conv_h16<false, 1>), grid, dim3(256), sh, st, p);
    return check_launch("k_conv_h16");
}

}  // namespace hl
