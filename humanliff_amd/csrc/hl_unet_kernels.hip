// Device kernels of the tri-plane UNet denoiser and the sampler update, MI355X (gfx950), fp32.
//
// What each kernel replaces in the reference (human_diffusion/improved_diffusion/):
//   k_conv_wino       nn.Conv2d 3x3 stride 1 on the large levels (ResBlock in_layers / out_layers, unet.py:149,164) by fused
//                     Winograd F(2x2,3x3), fp32.
//   k_conv_dma        every other nn.Conv2d 3x3 / 3x3 stride 2 / nearest-x2 + 3x3 / 1x1 / Conv1d k=1 (unet.py:68,100,149,
//                     164-184,237-239,378,486-518) as ONE implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32), operands
//                     staged by LDS-DMA; bias / residual / skip-sum fused into the store.  NHWC activations, channel
//                     pitch so concats are free.  k_conv_bf3: the same with bf16x3-emulated products (opt-in).
//   k_conv            the round's first kernel (register-staged, GroupNorm fused into the tile load): Cout <= 32, tiny layers.
//   k_gn_apply        GroupNorm-apply(+scale/shift)+SiLU materialised ahead of a DMA conv (nn.py:100, unet.py:212-216).
//   k_gn_partial/coef, k_gn_small  GroupNorm32(32, C) statistics (nn.py:17-19,100) folded to a per-(n,c) affine.
//   k_linear_small    time_embed / emb_layers nn.Linear at batch <= 8 (unet.py:151-157,366-370).
//   k_attention(_ks)  QKVAttention (unet.py:255-274): fp32 flash-style, softmax in fp32 (_ks: keys split over the waves).
//   k_prep_inputs     x.type(dtype), x + x_cond (unet.py:588,596) and NCHW -> NHWC.
#include "hl_unet_kernels.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "hl_stats.h"

namespace hl {
namespace {

// x*sigmoid(x) on raw v_exp_f32 (2^x) + v_rcp_f32, ~1 ulp each: 5 VALU ops.  exp2 overflowing to inf (v << 0)
// gives rcp(inf) = 0 -> -0, underflow (v >> 0) gives v; no fix-ups needed.
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v)); }

// ---------------------------------------------------------------------------------------------
// implicit-GEMM convolution
// ---------------------------------------------------------------------------------------------
// Order in which K is walked (and packed): groups of KG 16-channel chunks; inside a group all taps; inside a
// tap the group's chunks.  With KG = 2 a pixel's full 128-byte line (32 channels) is consumed by two consecutive
// k-tiles and the nine taps of a group (which re-read shifted copies of the same lines) are only 2*9 k-tiles
// apart, so they hit in the XCD's L2 - while the tap (and with it every padding predicate and the uniform
// address delta) changes only every KG k-tiles.
constexpr int KG = 4;
__host__ __device__ inline void kt_decode(int kt, int ncc, int taps, int &cc, int &tap) {
    const int per = KG * taps;
    const int cg = kt / per, r = kt - cg * per;
    const int g = (ncc - cg * KG) < KG ? (ncc - cg * KG) : KG;   // chunks in this group (last one may be short)
    // full groups come first, so r indexes inside this group only if cg is the short one at the end
    tap = r / g;
    cc = cg * KG + (r - tap * g);
}

// (struct ConvK - the kernel-side argument block of every convolution kernel - lives in hl_unet_kernels.h: k_conv_wino4w has its own file)

// (sum, sumsq) of a lane's values, combined over the two lane halves (same channel, other pixels): valid in every lane
template <int NV>
__device__ __forceinline__ void lane_stats(const float (&v)[NV], const bool (&ok)[NV], float &s, float &q) {
    s = 0.f; q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float x = ok[k] ? v[k] : 0.f;
        s += x;
        q += x * x;
    }
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
}
// ... added to the group totals of (img, group of view channel c0 + n)
template <int NV>
__device__ __forceinline__ void emit_stats(float *st, int N, long img, int c0, int cg, int n, long HW, int half, const float (&v)[NV], const bool (&ok)[NV]) {
    float s, q;
    lane_stats<NV>(v, ok, s, q);
    stat_add_run(st, N, img, (c0 + n) / cg, half == 0, HW, s, q);     // (call with the whole wave)
}

// MODE 0: raw input, 1: per-(n,c) affine (GroupNorm), 2: affine + SiLU.
// K order: see kt_decode.  Inside a tap every pointer just advances by 16 floats; a tap change recomputes the
// per-row pointers and padding predicates (set_tap).
template <int WM, int WN, int MT, int NT, int MODE, bool UPS>
__global__ __launch_bounds__(WM *WN * 64, (MT * NT <= 3) ? 4 : 2) void k_conv(const ConvK p) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32, NTHR = WM * WN * 64, LDA = 20;
    constexpr int A_F4 = BM * 4, B_F4 = BN * 4;
    constexpr int A_PER = (A_F4 + NTHR - 1) / NTHR, B_PER = (B_F4 + NTHR - 1) / NTHR;
    constexpr int STAGE = (BM + BN) * LDA;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][STAGE] tiles, then [nimg][2][Cin] GroupNorm affine

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // Workgroup -> (pixel tile, N block).  The dispatcher places workgroup b on XCD b % 8 (observed, speed only):
    // give every XCD a contiguous run of work items, N block fastest, so the N blocks of a pixel tile and the
    // vertically adjacent tiles (which re-read the same input rows for the 3x3 taps) share one XCD's L2.
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    const int mt_idx = wi / p.n_nblocks;
    const long m0 = (long)mt_idx * BM;
    const int n0 = (wi - mt_idx * p.n_nblocks) * BN;
    const int pad = p.ks >> 1;
    const int ncc = p.Cin >> 4;
    const int hw_out = p.Hout * p.Wout;
    const int img0 = (int)(m0 / hw_out);          // first image this pixel tile touches
    float *coef = lds + 2 * STAGE;

    // float4 element e of a tile -> (row, quarter): 8 consecutive lanes take 8 consecutive rows of one
    // quarter, which makes the ds_write_b128 groups bank-conflict free at a 20-dword row stride.
    int a_n[A_PER], a_y[A_PER], a_x[A_PER], a_q[A_PER], a_row[A_PER];
    bool a_in[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int e = tid + i * NTHR;
        a_row[i] = (e & 7) | ((e >> 5) << 3);
        a_q[i] = (e >> 3) & 3;
        const long P = m0 + a_row[i];
        a_in[i] = (e < A_F4) && (P < p.M);
        const long Pc = a_in[i] ? P : 0;
        a_n[i] = (int)(Pc / hw_out);
        const int rem = (int)(Pc - (long)a_n[i] * hw_out);
        a_y[i] = rem / p.Wout;
        a_x[i] = rem - a_y[i] * p.Wout;
    }
    const float *pA[A_PER];   // current tap, channel 0 of this thread's quarter
    int cofs[A_PER];          // LDS offset of this row's coefA quarter (coefB is Cin further)
    bool okA[A_PER];
    const float *pB[B_PER];
    int b_row[B_PER];
    bool b_in[B_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) cofs[i] = ((a_n[i] - img0) * 2) * p.Cin + a_q[i] * 4;

    const int nk_all = ncc * p.taps;
    const int kt0 = blockIdx.z * p.kt_per;
    const int nk = min(nk_all, kt0 + p.kt_per);
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int e = tid + i * NTHR;
        b_row[i] = (e & 7) | ((e >> 5) << 3);
        const int gn = n0 + b_row[i];
        b_in[i] = (e < B_F4) && gn < p.wrows;
        pB[i] = p.w + (long)(b_in[i] ? gn : 0) * p.Ktot + (long)kt0 * 16 + ((e >> 3) & 3) * 4;
    }

    // Without upsampling the input pixel of tap (ky,kx) is base + (ky*Win + kx) pixels for every row: the delta
    // is a wave-uniform scalar, the per-row state is one base offset and a 9-bit validity mask.
    long a_base[A_PER];
    unsigned a_mask[A_PER];
    if (!UPS) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int by = a_y[i] * p.stride - pad, bx = a_x[i] * p.stride - pad;
            a_base[i] = (((long)a_n[i] * p.Hin + by) * p.Win + bx) * p.in_pitch + a_q[i] * 4;
            unsigned m = 0;
            for (int t = 0; t < p.taps; ++t) {
                const int ky = (p.ks == 3) ? t / 3 : 0, kx = (p.ks == 3) ? t - ky * 3 : 0;
                const bool ok = a_in[i] && by + ky >= 0 && by + ky < p.Hin && bx + kx >= 0 && bx + kx < p.Win;
                m |= (ok ? 1u : 0u) << t;
            }
            a_mask[i] = m;
        }
    }
    auto set_tap = [&](int tap) {
        const int ky = (p.ks == 3) ? tap / 3 : 0, kx = (p.ks == 3) ? tap - ky * 3 : 0;
        if (!UPS) {
            const long delta = ((long)ky * p.Win + kx) * p.in_pitch;   // scalar
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const bool ok = (a_mask[i] >> tap) & 1u;
                okA[i] = ok;
                pA[i] = p.in + (ok ? a_base[i] + delta : (long)(a_q[i] * 4));   // padding: read pixel 0, zeroed later
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int vy = a_y[i] + ky - pad, vx = a_x[i] + kx - pad;
            const bool ok = a_in[i] && vy >= 0 && vy < 2 * p.Hin && vx >= 0 && vx < 2 * p.Win;
            const int iy = vy >> 1, ix = vx >> 1;
            okA[i] = ok;
            const long off = ok ? (((long)a_n[i] * p.Hin + iy) * p.Win + ix) * p.in_pitch : 0;
            pA[i] = p.in + off + a_q[i] * 4;
        }
    };
    int tap_l, cc_l;   // load cursor
    kt_decode(kt0, ncc, p.taps, cc_l, tap_l);
    int cg_l = cc_l / KG;                                        // current chunk group
    int gend_l = min(ncc, (cg_l + 1) * KG);                      // one past its last chunk
    set_tap(tap_l);

    f32x4 ra[A_PER], rb[B_PER];
    bool rv[A_PER];
    int rc = 0;   // channel offset of the staged tile

    auto load_tile = [&]() {
        const int c = cc_l * 16;
        rc = c;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            ra[i] = *reinterpret_cast<const f32x4 *>(pA[i] + c);
            rv[i] = okA[i];
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            rb[i] = *reinterpret_cast<const f32x4 *>(pB[i]);
            pB[i] += 16;
        }
        if (++cc_l == gend_l) {          // group's chunks done for this tap -> next tap, or next group
            if (++tap_l == p.taps) {
                tap_l = 0;
                ++cg_l;
                gend_l = min(ncc, (cg_l + 1) * KG);
            }
            cc_l = cg_l * KG;
            if (cc_l < ncc) set_tap(tap_l);
        }
    };
    auto store_tile = [&](int buf) {
        float *base = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            f32x4 v = ra[i];
            if (MODE != 0) {
                const f32x4 ca = *reinterpret_cast<const f32x4 *>(coef + cofs[i] + rc);
                const f32x4 cb = *reinterpret_cast<const f32x4 *>(coef + cofs[i] + p.Cin + rc);
                v = v * ca + cb;
            }
            if (MODE == 2) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
            if (!rv[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (A_F4 % NTHR == 0 || tid + i * NTHR < A_F4)
                *reinterpret_cast<f32x4 *>(base + a_row[i] * LDA + a_q[i] * 4) = v;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int e = tid + i * NTHR;
            if (B_F4 % NTHR == 0 || e < B_F4)
                *reinterpret_cast<f32x4 *>(base + (BM + b_row[i]) * LDA + ((e >> 3) & 3) * 4) = b_in[i] ? rb[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile();
    // the coefficient table (read by store_tile) is staged - or formed from the producers' totals - while the first tile's loads are in flight
    if (MODE != 0) {
        const long mlast = (m0 + BM - 1 < p.M ? m0 + BM - 1 : p.M - 1);
        const int nimg = (int)(mlast / hw_out) - img0 + 1;
        if (p.cA) {
            for (int e = tid; e < nimg * p.Cin; e += NTHR) {
                const int im = e / p.Cin, c = e - im * p.Cin;
                coef[(im * 2) * p.Cin + c] = p.cA[(long)(img0 + im) * p.Cin + c];
                coef[(im * 2 + 1) * p.Cin + c] = p.cB[(long)(img0 + im) * p.Cin + c];
            }
        } else {   // from the producers' group statistics (GnSrc): no coefficient launch in front of this kernel
            for (int im = 0; im < nimg; ++im)
                coef_to_lds(nullptr, nullptr, p.gn, p.N, img0 + im, coef + (im * 2) * p.Cin, coef + (im * 2 + 1) * p.Cin, coef + nimg * 2 * p.Cin, tid, NTHR);
        }
    }
    if (MODE != 0) __syncthreads();
    store_tile(0);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < nk) load_tile();
        const float *base = lds + buf * STAGE;
        f32x4 a[MT][2], b[NT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float *r = base + (wm * MT * 32 + i * 32 + (lane & 31)) * LDA + half * 8;
            a[i][0] = *reinterpret_cast<const f32x4 *>(r);
            a[i][1] = *reinterpret_cast<const f32x4 *>(r + 4);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float *r = base + (BM + wn * NT * 32 + j * 32 + (lane & 31)) * LDA + half * 8;
            b[j][0] = *reinterpret_cast<const f32x4 *>(r);
            b[j][1] = *reinterpret_cast<const f32x4 *>(r + 4);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s >> 2][s & 3], b[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane holds channel (lane&31) of pixels (r&3)+8*(r>>2)+4*half
    if (p.partial) {  // split-K: raw partial sums, finished by k_splitk_finish
        float *dst = p.partial + (long)blockIdx.z * p.M * p.Cout;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * NT * 32 + j * 32 + (lane & 31);
                if (n >= p.Cout) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long m = m0 + wm * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < p.M) dst[m * p.Cout + n] = acc[i][j][r];
                }
            }
        return;
    }
    const int hw = hw_out;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * NT * 32 + j * 32 + (lane & 31);
            if (n >= p.Cout) continue;
            const float bs = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wm * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bs;
                if (p.res) v += p.res[m * p.res_pitch + n];
                if (p.out_nchw) {
                    const long img = m / hw, rem = m - img * hw;
                    p.out[(img * p.Cout + n) * hw + rem] = v;
                } else {
                    p.out[m * p.out_pitch + n] = v;
                }
                if (p.out2) p.out2[m * p.out2_pitch + n] = v + p.res2[m * p.res2_pitch + n];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// k_conv_dma: the implicit GEMM of k_conv (K order of kt_decode, raw input) with both operand tiles staged by LDS-DMA
// (buffer_load_dwordx4 ... lds: no staging registers, no ds_write) into a ring of 3 stages, two tiles in flight,
// ONE raw s_barrier per k-tile and counted vmcnt waits.  WM waves of 32x96 stack along M: 128x96 (4 waves) or
// 256x96 (8 waves, half the weight staging per MAC).
// LDS rows are unpadded 64-byte lines (a DMA instruction fills 1 KiB linearly: 16 rows); bank conflicts of the
// ds_read_b128 fragment reads are avoided by XOR-swizzling the 16-byte quarter with (row>>2)&3, applied on the
// per-lane SOURCE offset of the DMA and on the read address.  Zero padding, ragged rows and the nearest-x2
// upsample (UPS) are all per-lane source offsets, recomputed only when the tap changes (every KG k-tiles).
// The GroupNorm(+SiLU) prologue cannot ride a DMA, so normalised inputs are materialised once by k_gn_apply.
// ---------------------------------------------------------------------------------------------

template <int WM, int NS, bool UPS>
__global__ __launch_bounds__(WM * 64, (WM == 8) ? 4 : 3) void k_conv_dma(const ConvK p) {
    static_assert(NS == 3, "the wait counts below assume a 3-stage ring");
    constexpr int BM = WM * 32, BN = 96, ROWS = BM + BN, STAGE_F = ROWS * 16;
    constexpr int NB2 = (WM == 4) ? 2 : 1;   // B instructions per wave (6 in total, instruction b = wave + WM*j)
    constexpr unsigned OOB = 0x80000000u;    // buffer offset past num_records (< 2 GiB): the load returns zeros
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    const int mt_idx = wi / p.n_nblocks;
    const long m0 = (long)mt_idx * BM;
    const int n0 = (wi - mt_idx * p.n_nblocks) * BN;
    const int pad = p.ks >> 1;
    const int ncc = p.Cin >> 4;
    const int hw_out = p.Hout * p.Wout;
    const unsigned pitch4 = (unsigned)p.in_pitch * 4u;

    // Both operands are fetched through buffer descriptors: address = base + per-lane offset (VGPR) + a wave-uniform
    // offset (SGPR), so walking K costs no vector ALU at all, and padding / ragged rows are lanes whose offset is
    // out of range (hardware returns zeros).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.in, (short)0, (int)((long)p.N * p.Hin * p.Win * p.in_pitch * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.w, (short)0, (int)((long)p.wrows * p.Ktot * 4), 0x00020000);

    // DMA instruction i of a tile fills rows 16i..16i+15: A rows by instructions wave + WM*j (j = 0,1), the 96 B rows
    // by instructions BM/16 + b with b = wave + WM*j < 6.
    // Lane L of an instruction writes physical quarter L&3 of row 16i + (L>>2).
    const int lrow = lane >> 2, pq = lane & 3;
    int iy0[2], ix0[2];    // top-left input coordinate of the 3x3 window (in the x2 grid when UPS); huge negative = no row
    unsigned nb[2];        // byte offset of image n, plus this lane's quarter
    unsigned a_cur[2];     // byte offset of this lane's 16 bytes for the current tap (or OOB)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave + WM * j) * 16 + lrow;
        const int ql = pq ^ ((row >> 2) & 3);
        const long P = m0 + row;
        const bool in = P < p.M;
        const long Pc = in ? P : 0;
        const int n = (int)(Pc / hw_out);
        const int rem = (int)(Pc - (long)n * hw_out);
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        iy0[j] = in ? oy * p.stride - pad : -(1 << 20);
        ix0[j] = ox * p.stride - pad;
        nb[j] = (unsigned)n * (unsigned)(p.Hin * p.Win) * pitch4 + ql * 16;
    }
    auto set_tap = [&](int tap) {
        const int ky = (p.ks == 3) ? tap / 3 : 0, kx = (p.ks == 3) ? tap - ky * 3 : 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int iy = iy0[j] + ky, ix = ix0[j] + kx;
            const int Hv = UPS ? 2 * p.Hin : p.Hin, Wv = UPS ? 2 * p.Win : p.Win;
            const bool ok = iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
            const int sy = UPS ? iy >> 1 : iy, sx = UPS ? ix >> 1 : ix;
            a_cur[j] = ok ? nb[j] + (unsigned)(sy * p.Win + sx) * pitch4 : OOB;
        }
    };
    const int nk_all = ncc * p.taps;
    const int kt0 = blockIdx.z * p.kt_per;
    const int nk = min(nk_all, kt0 + p.kt_per);
    unsigned b_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int brow = (wave + WM * j) * 16 + lrow;         // valid < 96
        const int ql = pq ^ (((BM + brow) >> 2) & 3);
        const int gn = n0 + brow;
        b_voff[j] = (brow < BN && gn < p.wrows) ? (unsigned)gn * (unsigned)p.Ktot * 4u + ql * 16 : OOB;
    }
    const bool has_b0 = wave < 6;                              // B instruction wave
    const bool has_b1 = (NB2 == 2) && wave + WM < 6;           // B instruction wave + WM
    const int n_w = 2 + (has_b0 ? 1 : 0) + (has_b1 ? 1 : 0);   // DMA instructions of this wave per tile

    int tap_i, cc_i;                                           // issue cursor
    kt_decode(kt0, ncc, p.taps, cc_i, tap_i);
    int cg_i = cc_i / KG, gend_i = min(ncc, (cg_i + 1) * KG);
    int soffB = kt0 * 64;
    set_tap(tap_i);
    auto issue = [&](int stage) {
        float *dst = lds + stage * STAGE_F + wave * 256;       // + WM*256 floats per j (WM instructions further)
        const int soffA = cc_i * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(dst + j * (WM * 256)), 16,
                                                     a_cur[j], soffA, 0, 0);
        if (has_b0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(dst + BM * 16), 16,
                                                     b_voff[0], soffB, 0, 0);
        if (has_b1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(dst + BM * 16 + WM * 256),
                                                     16, b_voff[1], soffB, 0, 0);
        soffB += 64;
        if (++cc_i == gend_i) {
            if (++tap_i == p.taps) { tap_i = 0; ++cg_i; gend_i = min(ncc, (cg_i + 1) * KG); }
            cc_i = cg_i * KG;
            set_tap(tap_i);
        }
    };
    auto wait_younger = [&]() {   // all but this wave's DMAs of the youngest tile have landed
        if (n_w == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n_w == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    };

    // fragment read offsets (floats) inside a stage; the three B fragments sit 32 rows (512 floats) apart with the
    // same swizzle phase, so they share one address register and differ by immediate offsets
    const int ra = wave * 32 + (lane & 31);
    const int sa = (ra >> 2) & 3;
    const int a_off0 = ra * 16 + (((2 * half) ^ sa) << 2), a_off1 = ra * 16 + (((2 * half + 1) ^ sa) << 2);
    const int rb = BM + (lane & 31);
    const int sb = (rb >> 2) & 3;
    const int b_off0 = rb * 16 + (((2 * half) ^ sb) << 2), b_off1 = rb * 16 + (((2 * half + 1) ^ sb) << 2);

    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // Software pipeline over HALF k-tiles (8 of the 16 channels = 12 MFMAs): the LDS reads of the next half are in
    // flight while the MFMAs of the current half run, and the per-tile barrier is followed directly by MFMAs whose
    // operands are already in registers.  Per tile t (stage t % 3, a compile-time constant in the unrolled body):
    //     3 mfma h0(t) | read h1(t) | 9 mfma h0(t) | wait DMA(t+1), barrier | 3 mfma h1(t) | issue DMA(t+3), read h0(t+1) | 9 mfma h1(t)
    const int ntiles = nk - kt0;
    f32x4 a0, a1, b0[3], b1[3];
    if (ntiles > 0) {
        issue(0);
        if (ntiles > 1) { issue(1); wait_younger(); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ntiles > 2) issue(2);
        a0 = *reinterpret_cast<const f32x4 *>(lds + a_off0);
#pragma unroll
        for (int j = 0; j < 3; ++j) b0[j] = *reinterpret_cast<const f32x4 *>(lds + b_off0 + j * 512);
    }
    auto body = [&](auto uc, int t) {
        constexpr int U = decltype(uc)::value, UN = (U + 1) % NS;
        const float *base = lds + U * STAGE_F, *nbase = lds + UN * STAGE_F;
        // (the reads go AFTER the first MFMAs in program order: the compiler's wait for h0 is a full lgkmcnt(0))
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0[j][0], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a1 = *reinterpret_cast<const f32x4 *>(base + a_off1);
#pragma unroll
        for (int j = 0; j < 3; ++j) b1[j] = *reinterpret_cast<const f32x4 *>(base + b_off1 + j * 512);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 1; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[j][s], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const bool more = t + 1 < ntiles;
        if (more) {
            if (t + 2 < ntiles) wait_younger(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every wave is done reading tile t before its stage refills
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], b1[j][0], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            if (t + NS < ntiles) issue(U);
            a0 = *reinterpret_cast<const f32x4 *>(nbase + a_off0);
#pragma unroll
            for (int j = 0; j < 3; ++j) b0[j] = *reinterpret_cast<const f32x4 *>(nbase + b_off0 + j * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 1; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[j][s], acc[j], 0, 0, 0);
    };
    for (int t = 0; t < ntiles; t += NS) {
        body(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) body(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < ntiles) body(std::integral_constant<int, 2>{}, t + 2);
    }

    // epilogue (same contract as k_conv)
    if (p.partial) {
        float *dst = p.partial + (long)blockIdx.z * p.M * p.Cout;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = n0 + j * 32 + (lane & 31);
            if (n >= p.Cout) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < p.M) dst[m * p.Cout + n] = acc[j][r];
            }
        }
        return;
    }
    // GroupNorm statistics of the stored tile: when the tile's WM*32 rows lie in one image the waves' sums meet in LDS (the staging ring
    // is free now) and the workgroup adds ONE pair per channel to the totals; otherwise (low levels) every wave adds its 32 rows' pair
    const bool wg_stats = (p.st1 || p.st2) && hw_out % (WM * 32) == 0;
    float *red1 = lds, *red2 = lds + 2 * WM * 96;
    unsigned long long *lgrp = reinterpret_cast<unsigned long long *>(lds + 4 * WM * 96);
    if (wg_stats) {                                       // (slower waves may still be reading the last stage; no LDS-DMA piece may still be in flight)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // per 32-column block: all loads (residual, second residual) are issued before any store, so they overlap
    // instead of serialising behind the stores (res may alias out)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int n = n0 + j * 32 + (lane & 31);
        if (n >= p.Cout) continue;
        const float bs = p.bias ? p.bias[n] : 0.f;
        const long mb = m0 + wave * 32 + 4 * half;
        float v[16], v2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[j][r] + bs;
        if (p.res) {
            float rr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = mb + (r & 3) + 8 * (r >> 2);
                rr[r] = m < p.M ? p.res[m * p.res_pitch + n] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += rr[r];
        }
        if (p.out2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = mb + (r & 3) + 8 * (r >> 2);
                v2[r] = m < p.M ? p.res2[m * p.res2_pitch + n] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v2[r] += v[r];
        }
        if (wg_stats) {
            bool okr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) okr[r] = mb + (r & 3) + 8 * (r >> 2) < p.M;
            float s_, q_;
            if (p.st1) { lane_stats<16>(v, okr, s_, q_); if (half == 0) wg_stat_put<96>(red1, wave, j * 32 + (lane & 31), s_, q_); }
            if (p.st2) { lane_stats<16>(v2, okr, s_, q_); if (half == 0) wg_stat_put<96>(red2, wave, j * 32 + (lane & 31), s_, q_); }
        } else if ((p.st1 || p.st2) && m0 + wave * 32 < p.M) {   // this wave's 32 rows of M (the launcher checked: they lie in one image)
            bool okr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) okr[r] = mb + (r & 3) + 8 * (r >> 2) < p.M;
            if (p.st1) emit_stats<16>(p.st1, p.N, (m0 + wave * 32) / hw_out, p.st1_c0, p.st1_cg, n, hw_out, half, v, okr);
            if (p.st2) emit_stats<16>(p.st2, p.N, (m0 + wave * 32) / hw_out, p.st2_c0, p.st2_cg, n, hw_out, half, v2, okr);
        }
        if (p.out_nchw) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = mb + (r & 3) + 8 * (r >> 2);
                const long img = m / hw_out, rem = m - img * hw_out;
                if (m < p.M) p.out[(img * p.Cout + n) * hw_out + rem] = v[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = mb + (r & 3) + 8 * (r >> 2);
                if (m < p.M) p.out[m * p.out_pitch + n] = v[r];
            }
        }
        if (p.out2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = mb + (r & 3) + 8 * (r >> 2);
                if (m < p.M) p.out2[m * p.out2_pitch + n] = v2[r];
            }
        }
    }
    if (wg_stats) {
        __syncthreads();
        if (p.st1) wg_group_flush<96, WM>(red1, lgrp, p.st1, p.N, m0 / hw_out, p.Cout, n0, p.st1_c0, p.st1_cg, hw_out, tid);
        if (p.st2) wg_group_flush<96, WM>(red2, lgrp, p.st2, p.N, m0 / hw_out, p.Cout, n0, p.st2_c0, p.st2_cg, hw_out, tid);
    }
}

// ---------------------------------------------------------------------------------------------
// k_conv_wino: 3x3 / stride 1 convolution by Winograd F(2x2, 3x3) in fp32, fused in one kernel
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        16 multiplies per 2x2 outputs instead of 36 (2.25x fewer MFMAs)
// A workgroup owns 8x8 output tiles of 2x2 pixels (a 16x16 pixel block) x 64 output channels x all 16 frequencies.
// Wave w = 4g + i: tile group g (rows of tiles 4g..4g+3, 32 tiles = the MFMA M dimension) and frequency ROW i: it
// keeps M[i][0..3] for 2 x 32 output channels = 8 accumulator tiles (128 registers).  Per k-tile of 8 input channels:
//   * the 18x18 input patch and the 16 pre-transformed 64x8 weight slices U = G g G^T arrive by LDS-DMA (3-stage
//     ring, buffer addressing, per-lane offsets fixed for the whole K walk; the zero padding is out-of-range lanes);
//   * a lane (tile, channel-half) reads the two patch rows its frequency row needs (B^T row i), forms
//     V[i][0..3] = (row transform) (column transform) with 32 adds, and issues 32 MFMAs against the weight slices.
// After the K walk every wave applies the column half of A^T . A to its accumulators, the four frequency rows meet in
// LDS, and each wave finishes one (column parity, channel block) of the outputs with the usual epilogue.
// LDS layouts are chosen so that the 8 lanes a ds_read_b128 serves per cycle read consecutive 16-byte chunks:
//   patch chunk  (((row*2 + (col&1))*10 + (col>>1))*2 + (half ^ bit 1 of row)   (4 channels of one pixel: the two channel halves of a
//                pixel are neighbours, fetched by neighbouring DMA lanes - 32 contiguous bytes, one L2 request instead of two, as in
//                k_conv_wino4; the swap by the row bit puts the tiles of rows ty and ty+1 on the even / odd 16-byte slots, so the
//                lane groups of a ds_read_b128 still hit 16 different slots - checked by enumeration)
//   weight chunk ((f*2 + ct)*2 + half)*32 + cout                   (4 channels of one output channel)
// ---------------------------------------------------------------------------------------------
template <bool UPS>
__global__ __launch_bounds__(256, 2) void k_conv_wino(const ConvK p) {
#if __HIP_DEVICE_COMPILE__   // device pass only (see k_conv_bf3)
    // One tile group of 32 tiles (4 rows of 8 tiles = 16 x 8 pixels) per workgroup, 4 waves (frequency rows), 2-stage ring, two
    // independent workgroups per CU (their prologues / epilogues / barriers overlap).  (Round 1 also carried a two-group 8-wave
    // variant and one with the channel blocks on separate waves; both measured slower - 249 / 237 vs 261 TFLOP/s - and are gone.)
    // (Round 2 tried a variant that takes the RAW tensor and applies the consuming layer's GroupNorm affine + SiLU to the patch in LDS,
    //  in place, by the wave that DMA'd it - no k_gn_apply pass.  It was 25-30 % slower per launch (760 -> 995 us at 256x256): the
    //  8 x (fma, exp, rcp, 2 mul) per lane and k-tile plus the coefficient fetch sit between the patch DMA and the barrier that
    //  publishes it, whichever way they were placed; the separate pass costs 6 % of the step and mostly overlaps the other stream.)
    constexpr int G = 1, CW = 1;
    constexpr int NW = 4 * G * CW, NCT = 2 / CW, PR = 8 * G + 2;   // waves, channel blocks per wave, patch rows (2-stage ring)
    constexpr int U_F = 16 * 2 * 2 * 32 * 4, P_REAL = 2 * PR * 2 * 10, NP = (P_REAL + 63) / 64, P_F = NP * 256, STAGE_F = U_F + P_F;
    constexpr int NU = 32 / NW, NTOT = 32 + NP, NJ = (NTOT + NW - 1) / NW;   // weight instructions per wave, all, rounds
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = wave & 3, ctw = (CW == 2) ? ((wave >> 2) & 1) : 0, g = wave >> (CW == 2 ? 3 : 2);   // frequency row, channel block, tile group
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    const int tb = wi / p.n_nblocks, nb = wi - tb * p.n_nblocks;
    const int n0 = nb * 64;
    // UPS: the convolution runs on the nearest-x2 upsampled image (unet.py:77-79); only the patch DMA knows - it fetches source
    // pixel (y>>1, x>>1) - so the upsampled tensor is never materialised
    const int Hv = UPS ? 2 * p.Hin : p.Hin, Wv = UPS ? 2 * p.Win : p.Win;
    const int bw = Wv >> 4, bh = Hv / (8 * G);
    const int img = tb / (bw * bh), brem = tb - img * (bw * bh);
    const int y0 = (brem / bw) * (8 * G), x0 = (brem - (brem / bw) * bw) * 16;
    const int nkt = p.Cin >> 3;
    const unsigned pitch4 = (unsigned)p.in_pitch * 4u;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.in, (short)0, (int)((long)p.N * p.Hin * p.Win * p.in_pitch * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.w_wino, (short)0, (int)((long)p.n_nblocks * nkt * U_F * 4), 0x00020000);

    // DMA instructions of a k-tile: 32 weight instructions (plain 1 KiB copies) then NP patch instructions; instruction
    // q = wave + NW*j: the first NU rounds are weights, the rest patch instructions q - 32 (< NP)
    unsigned pv[NJ - NU];
    int n_p = 0;
#pragma unroll
    for (int j = 0; j < NJ - NU; ++j) {
        const int pi = wave + NW * (NU + j) - 32;
        const int c = pi * 64 + lane;                             // patch chunk
        const int hs = c & 1, tq = c >> 1, jc = tq % 10, t1 = tq / 10, par = t1 & 1, row = t1 >> 1, h = hs ^ ((row >> 1) & 1);
        const int y = y0 - 1 + row, x = x0 - 1 + 2 * jc + par;
        const bool ok = pi < NP && c < P_REAL && jc < 9 && y >= 0 && y < Hv && x >= 0 && x < Wv;
        const int ys = UPS ? y >> 1 : y, xs = UPS ? x >> 1 : x;
        pv[j] = ok ? (unsigned)((img * p.Hin + ys) * p.Win + xs) * pitch4 + h * 16 : OOB;
        n_p += (pi < NP) ? 1 : 0;
    }
    const int kt0 = blockIdx.z * p.kt_per;                        // split-K over input channels: this slab's k-tiles
    int soffA = kt0 * 32;
    static_assert(NJ - NU <= 2, "at most two patch instructions per wave");

    // patch read offsets (floats): tile T = lane & 31 -> (ty, tx); rows rA, rB of B^T row fi; column c: parity c&1, + (c>>1)
    const int T = lane & 31, ty = 4 * g + (T >> 3), tx = T & 7;
    const int rA = (fi == 0) ? 0 : (fi == 2 ? 2 : 1), rB = (fi == 0) ? 2 : (fi == 1 ? 2 : (fi == 2 ? 1 : 3));
    const float sB = (fi == 1) ? 1.f : -1.f;                      // T = d[rA] + sB * d[rB]
    const int pA = U_F + (((((2 * ty + rA) * 2) * 10 + tx) * 2) + (half ^ (((2 * ty + rA) >> 1) & 1))) * 4;
    const int pB = U_F + (((((2 * ty + rB) * 2) * 10 + tx) * 2) + (half ^ (((2 * ty + rB) >> 1) & 1))) * 4;
    // column c of a row: + ((c & 1) * 10 + (c >> 1)) * 8 floats
    const int u_off = ((fi * 4 * 2) * 2 + half) * 32 * 4 + (lane & 31) * 4;   // + ((f' * 2 + ct) * 2) * 128 floats

    f32x16 acc[4][NCT];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][c][r] = 0.f;

    // Pipeline in units of (frequency f', channel block ct) = 4 MFMAs: the weight chunk of the next unit is read while the
    // current one multiplies; the barrier for k-tile t+1 sits in front of unit 6, followed by the patch reads of t+1,
    // whose transform is computed behind the last MFMAs of tile t.
    const int ntiles = min(nkt, kt0 + p.kt_per) - kt0;
    auto read_patch = [&](const float *base, f32x4(&da)[4], f32x4(&db)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int co = ((c & 1) * 10 + (c >> 1)) * 8;
            da[c] = *reinterpret_cast<const f32x4 *>(base + pA + co);
            db[c] = *reinterpret_cast<const f32x4 *>(base + pB + co);
        }
    };
    auto transform = [&](const f32x4(&da)[4], const f32x4(&db)[4], f32x4(&V)[4]) {
        f32x4 tc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) tc[c] = da[c] + sB * db[c];
        V[0] = tc[0] - tc[2];
        V[1] = tc[1] + tc[2];
        V[2] = tc[2] - tc[1];
        V[3] = tc[1] - tc[3];
    };
    f32x4 V[4], ub[2][NCT], da[4], db[4];   // ub[buffer][ct]
    auto read_u = [&](const float *base, int f, f32x4(&u2)[NCT]) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) u2[c] = *reinterpret_cast<const f32x4 *>(base + u_off + ((f * 2 + (CW == 2 ? ctw : c)) * 2) * 128);
    };
    {
        // The four waves of a workgroup use DISJOINT weight slices (their own frequency row), so each wave
        // DMAs exactly its own 8 chunks per k-tile and nobody else reads them: no barrier is needed for the weights, and the
        // refill of a chunk pair is issued right behind the MFMAs that consumed it - two DMA instructions per unit instead of
        // a burst of ten behind the barrier, each with two k-tiles to land.  Only the patch (1-2 instructions per wave) stays
        // behind the barrier.  Per wave the VMEM queue therefore carries, per k-tile, the fixed sequence
        //     U(.,0) U(.,1) U(.,2) P(.) U(.,3)            (2, 2, 2, n_p, 2 instructions)
        // for the tile two ahead, and every wait below is an exact count of the younger instructions in that queue.
        const unsigned uvp = (unsigned)lane * 16u;
        int soffP = (nb * nkt + kt0) * (U_F * 4) + (8 * fi) * 1024;   // this wave's first chunk of k-tile kt0
        auto issue_u = [&](int stage, int f, int soff) {             // both channel blocks of frequency f'
#pragma unroll
            for (int c = 0; c < 2; ++c)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsU, (__attribute__((address_space(3))) void *)(lds + stage * STAGE_F + (8 * fi + 2 * f + c) * 256), 16, uvp,
                    soff + (2 * f + c) * 1024, 0, 0);
        };
        auto issue_p = [&](int stage) {
            float *dp = lds + stage * STAGE_F + U_F + wave * 256;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)dp, 16, pv[0], soffA, 0, 0);
            if (n_p > 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(dp + NW * 256), 16, pv[1],
                                                         soffA, 0, 0);
            soffA += 32;
        };
        auto issue_tile = [&](int stage) {   // the whole sequence at once (prologue only)
            issue_u(stage, 0, soffP); issue_u(stage, 1, soffP); issue_u(stage, 2, soffP);
            issue_p(stage);
            issue_u(stage, 3, soffP);
            soffP += U_F * 4;
        };
        auto wait_vm = [&](int n1, int n2) {   // s_waitcnt vmcnt(n_p == 1 ? n1 : n2)
            const int n = n_p > 1 ? n2 : n1;
            if (n == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (n == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else if (n == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
            else if (n == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (n == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        if (ntiles > 0) {
            issue_tile(0);
            if (ntiles > 1) { issue_tile(1); wait_vm(11, 12); }           // younger than P(0): U(0,3) + the 8 + n_p of tile 1
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");         // younger than P(0): U(0,3)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            read_patch(lds, da, db);
            read_u(lds, 0, ub[0]);
            transform(da, db, V);
        }
        auto bodyp = [&](auto uc, int t) {
            constexpr int U = decltype(uc)::value, UN = U ^ 1;
            const float *base = lds + U * STAGE_F, *nbase = lds + UN * STAGE_F;
            const bool more = t + 1 < ntiles, more2 = t + 2 < ntiles;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[f][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[f][0], ub[f & 1][c][0], acc[f][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (f < 3) {
                    // U(t, f+1) was issued two tiles ago; younger: the rest of that tile's sequence, the 8 + n_p of tile t-1's and
                    // the 2f of this tile's issues so far.  The last two tiles (no refills any more) simply drain the queue.
                    if (!more2) { if (f == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                    else if (f == 0) wait_vm(14, 16);      // 4 + n_p + 8 + n_p
                    else if (f == 1) wait_vm(14, 16);      // 2 + n_p + 8 + n_p + 2
                    else wait_vm(13, 14);                  // 8 + n_p + 4
                    read_u(base, f + 1, ub[(f + 1) & 1]);
                } else if (more) {
                    // P(t+1) (and the older U(t+1,0)) have landed once only U(t+1,3) and this tile's three refills are younger
                    if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();            // patch of tile t+1 visible to all; every wave has consumed patch t
                    asm volatile("" ::: "memory");
                    if (more2) issue_p(U);
                    read_patch(nbase, da, db);
                    read_u(nbase, 0, ub[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 1; s < 4; ++s)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[f][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[f][s], ub[f & 1][c][s], acc[f][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more2) issue_u(U, f, soffP);              // the chunk pair just consumed refills for tile t+2
            }
            if (more2) soffP += U_F * 4;
            if (more) transform(da, db, V);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        for (int t = 0; t < ntiles; t += 2) {
            bodyp(std::integral_constant<int, 0>{}, t);
            if (t + 1 < ntiles) bodyp(std::integral_constant<int, 1>{}, t + 1);
        }
    }

    // column half of the output transform, then the four frequency rows meet in LDS
    __syncthreads();
    constexpr int EXW = 2 * NCT * 16 * 64;                        // floats per wave: [b][ct][r][lane]
    float *ex = lds + wave * EXW;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = acc[0][c][r], m1 = acc[1][c][r], m2 = acc[2][c][r], m3 = acc[3][c][r];
            ex[((0 * NCT + c) * 16 + r) * 64 + lane] = (m0 + m1) + m2;
            ex[((1 * NCT + c) * 16 + r) * 64 + lane] = (m1 - m2) - m3;
        }
    __syncthreads();
    // CW = 1: wave (g, fi) finishes column parity b = fi >> 1, channel block ct = fi & 1 of its tile group (32 outputs per lane);
    // CW = 2: wave (g, ctw, fi) finishes b = fi & 1 of its own channel block for accumulator rows 8*(fi>>1).. (16 outputs per lane).
    // All loads (residuals) are issued before any store, so they overlap instead of serialising behind the stores
    // (res may alias out).
    constexpr int NR = 16 / CW, NO = 2 * NR;
    const int b = (CW == 2) ? (fi & 1) : (fi >> 1), ct = (CW == 2) ? ctw : (fi & 1), r0 = (CW == 2) ? 8 * (fi >> 1) : 0;
    const int n = n0 + ct * 32 + (lane & 31);
    const float bs = (p.bias && !p.partial) ? p.bias[n] : 0.f;
    const long hw = (long)Hv * Wv;
    float v[NO];
    long mm[NO];
    const float *zbase = lds + ((CW == 2 ? (g * 2 + ctw) : g) * 4) * EXW + ((b * NCT + (CW == 2 ? 0 : ct)) * 16) * 64 + lane;
#pragma unroll
    for (int rr_ = 0; rr_ < NR; ++rr_) {
        const int r = r0 + rr_;
        const float *zz = zbase + r * 64;
        const float z0 = zz[0], z1 = zz[EXW], z2 = zz[2 * EXW], z3 = zz[3 * EXW];
        const int Tr = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int oy = y0 + 2 * (4 * g + (Tr >> 3)), ox = x0 + 2 * (Tr & 7) + b;
        v[2 * rr_] = ((z0 + z1) + z2) + bs;
        v[2 * rr_ + 1] = ((z1 - z2) - z3) + bs;
        mm[2 * rr_] = ((long)img * Hv + oy) * Wv + ox;
        mm[2 * rr_ + 1] = mm[2 * rr_] + Wv;
    }
    if (p.partial) {   // split-K: the output transform is linear, so slabs are summed in the output domain by k_splitk_finish
        float *dst = p.partial + (long)blockIdx.z * p.M * p.Cout;
#pragma unroll
        for (int k = 0; k < NO; ++k) dst[mm[k] * p.Cout + n] = v[k];
        return;
    }
    if (p.res) {
        float rr[NO];
#pragma unroll
        for (int k = 0; k < NO; ++k) rr[k] = p.res[mm[k] * p.res_pitch + n];
#pragma unroll
        for (int k = 0; k < NO; ++k) v[k] += rr[k];
    }
    float v2[NO];
    if (p.out2) {
#pragma unroll
        for (int k = 0; k < NO; ++k) v2[k] = p.res2[mm[k] * p.res2_pitch + n];
#pragma unroll
        for (int k = 0; k < NO; ++k) v2[k] += v[k];
    }
    if (p.st1 || p.st2) {   // (tile block, column parity): 64 pixels of one image
        bool all[NO];
#pragma unroll
        for (int k = 0; k < NO; ++k) all[k] = true;
        if (p.st1) emit_stats<NO>(p.st1, p.N, img, p.st1_c0, p.st1_cg, n, hw, half, v, all);
        if (p.st2) emit_stats<NO>(p.st2, p.N, img, p.st2_c0, p.st2_cg, n, hw, half, v2, all);
    }
    if (p.out_nchw) {
#pragma unroll
        for (int k = 0; k < NO; ++k) p.out[((long)img * p.Cout + n) * hw + (mm[k] - (long)img * hw)] = v[k];
    } else {
#pragma unroll
        for (int k = 0; k < NO; ++k) p.out[mm[k] * p.out_pitch + n] = v[k];
    }
    if (p.out2) {
#pragma unroll
        for (int k = 0; k < NO; ++k) p.out2[mm[k] * p.out2_pitch + n] = v2[k];
    }
#endif
}

// weights -> U = G g G^T per (cout, cin), laid out as k_conv_wino stages them: [cout/64][cin/8][f][ct][half][32][4]
// One thread per (output channel, input channel): the nine taps are read once and all 16 frequencies written (the first version had
// a thread per OUTPUT element - every tap gathered 16 times at a 36-byte lane stride: 55 us per layer on average, 10 ms of a
// training step, which re-lays every weight twice).  A block = 32 output channels x 8 input channels = one k-tile of one channel
// half; per frequency it writes two runs of 512 contiguous bytes.  The sum is formed in the same order as before (bit-equal copy).
__global__ __launch_bounds__(256) void k_pack_conv_wino(const float *__restrict__ w, int Cout, int Cin, int Cin_pad, float *__restrict__ dst, int tf) {
    const int nkt = Cin_pad >> 3;
    const int t = threadIdx.x, s = t & 3, nn = (t >> 2) & 31, hf = t >> 7;
    const long nblk = (long)(Cout >> 5) * nkt;
    for (long bi = blockIdx.x; bi < nblk; bi += gridDim.x) {
        const int kt = (int)(bi % nkt), cb = (int)(bi / nkt), nb = cb >> 1, ct = cb & 1;
        const int co = cb * 32 + nn, ci = kt * 8 + hf * 4 + s;
        const bool ok = ci < Cin && co < Cout;
        double g[9];
        if (ok) {
            const float *q = tf ? w + ((long)ci * Cout + co) * 9 : w + ((long)co * Cin + ci) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) g[k] = (double)q[tf ? 8 - k : k];
        }
        const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        float *o = dst + ((((long)nb * nkt + kt) * 16 * 2 + ct) * 2 + hf) * 128 + nn * 4 + s;
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            const int i = f >> 2, j = f & 3;
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int vv = 0; vv < 3; ++vv) acc += G[i][u] * g[u * 3 + vv] * G[j][vv];
            o[(long)f * 512] = ok ? (float)acc : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_conv_wino4: 3x3 / stride 1 convolution by Winograd F(4x4, 3x3) in fp32, fused in one kernel
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        36 multiplies per 4x4 outputs instead of 144 (4x fewer MFMAs than direct,
//                                                  1.78x fewer than F(2x2,3x3))
// Interpolation points (0, +a, -a, +b, -b, inf) with a = 3/4, b = 3/2 instead of the textbook (0, +-1, +-2, inf): every entry of
// B^T and A^T is still exact in fp32, and the error against float64 on a 192-channel layer is 3x the F(2x2) kernel's instead of
// 6x (rms 7e-7 on outputs of std 0.6; scripts/wino_points.py).  Rows of B^T (each with leading coefficient 1):
//     0   : a2b2 d0 - (a2+b2) d2 + d4                       +-a : (-b2 d2 + d4) +- a (-b2 d1 + d3)
//     inf : a2b2 d1 - (a2+b2) d3 + d5                       +-b : (-a2 d2 + d4) +- b (-a2 d1 + d3)
// A workgroup owns 8x4 tiles of 4x4 outputs (a 32x16 pixel block, 32 tiles = the MFMA M dimension) x 32 output channels x all 36
// frequencies; wave (fr, fc) of its 4 waves owns the 3x3 frequency block rows {0,+a,-a} or {+b,-b,inf} x columns likewise:
// 9 accumulator tiles = 144 registers, 2 waves / SIMD, two workgroups per CU (2 x 76.5 KB of LDS).
// Per k-tile of 8 input channels:
//   * the 34x18 input patch arrives by LDS-DMA into a 2-stage ring (22 instructions per workgroup) and the wave's nine 32x8
//     weight slices U = G g G^T by LDS-DMA into nine PRIVATE single-buffered 1 KiB slots: a slot is refilled for the next k-tile
//     right behind the MFMAs that consumed it, so the weights need no barrier and every wait is an exact vmcnt over the fixed
//     per-wave instruction sequence  P(t+2) U(t+1,0) ... U(t+1,8);
//   * transform phase: a lane (tile, channel half) reads the 5x5 window positions its frequency block needs (a block uses rows /
//     columns 0..4 or 1..5 of the 6x6 window), 25 ds_read_b128, and forms V[3][3] with 48 fused multiply-adds per channel;
//   * MFMA phase: 36 MFMAs against the weight slots.  The two phases of a wave do not overlap (no registers for a second V); the
//     other workgroup's wave on the same SIMD fills the matrix pipe meanwhile.
// After the K walk the 36 frequencies of a (tile, channel) meet in LDS (two rounds of 16 tiles, 72 KB each) and every thread applies
// A^T . A to two tiles per round and finishes their 4x4 pixels with the usual epilogue.
// Patch chunk (4 channels of one pixel): (row*36 + (col&3)*9 + (col>>2))*2 + (half ^ bit 2 of row): the two halves of a pixel are
// neighbours, fetched by neighbouring DMA lanes - 32 contiguous bytes, one L2 request instead of two (with one 16-byte piece per
// request the kernel ran at the L2's request rate: 0.6 requests per clock and channel, 94 % hits) - and the swap by the row bit puts
// the tiles of rows ty and ty+1 on the even and the odd 16-byte slots, so that every 16-lane group of a ds_read_b128 still hits 16
// different slots.  Weight slot: [half][32 couts][4 channels] = the lane order of the DMA.
// ---------------------------------------------------------------------------------------------
constexpr float W4_A = 0.75f, W4_B = 1.5f;
constexpr float W4_C0 = W4_A * W4_A * W4_B * W4_B, W4_C2 = -(W4_A * W4_A + W4_B * W4_B);
__device__ __forceinline__ f32x4 fma4(float c, f32x4 x, f32x4 y) { return __builtin_elementwise_fma((f32x4)(c), x, y); }
// three rows of B^T applied to five consecutive window elements: BLK 0 = rows (0, +a, -a) on d0..d4, BLK 1 = rows (+b, -b, inf) on d1..d5
template <int BLK>
__device__ __forceinline__ void w4_fwd(const f32x4 (&x)[5], f32x4 (&o)[3]) {
    if constexpr (BLK == 0) {
        o[0] = fma4(W4_C0, x[0], fma4(W4_C2, x[2], x[4]));
        const f32x4 e = fma4(-W4_B * W4_B, x[2], x[4]), t = fma4(-W4_B * W4_B, x[1], x[3]);
        o[1] = fma4(W4_A, t, e);
        o[2] = fma4(-W4_A, t, e);
    } else {
        const f32x4 e = fma4(-W4_A * W4_A, x[1], x[3]), t = fma4(-W4_A * W4_A, x[0], x[2]);
        o[0] = fma4(W4_B, t, e);
        o[1] = fma4(-W4_B, t, e);
        o[2] = fma4(W4_C0, x[0], fma4(W4_C2, x[2], x[4]));
    }
}
// one side of the output transform: four outputs from the six frequencies (0, +a, -a, +b, -b, inf)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(float c, f32x2 x, f32x2 y) { return __builtin_elementwise_fma((f32x2)(c), x, y); }
__device__ __forceinline__ void w4_out(f32x2 m0, f32x2 m1, f32x2 m2, f32x2 m3, f32x2 m4, f32x2 m5, f32x2 (&y)[4]) {   // two channels at once
    const f32x2 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y[0] = (m0 + s1) + s2;
    y[1] = fma2(W4_B, d2, W4_A * d1);
    y[2] = fma2(W4_B * W4_B, s2, (W4_A * W4_A) * s1);
    y[3] = fma2(W4_B * W4_B * W4_B, d2, fma2(W4_A * W4_A * W4_A, d1, m5));
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool UPS, bool BLK>
__global__ __launch_bounds__(256, 2) void k_conv_wino4(const ConvK p) {
#if __HIP_DEVICE_COMPILE__   // device pass only (see k_conv_bf3)
    constexpr int PRW = 36, PROWS = 18, P_REAL = 2 * PROWS * PRW, NP = (P_REAL + 63) / 64, P_F = P_REAL * 4;   // patch: pixels / row, rows, chunks, DMAs, floats
    constexpr int U_F = 36 * 256;                                                                            // 36 weight slots of 1 KiB
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    const int tb = wi / p.n_nblocks, nb = wi - tb * p.n_nblocks;
    const int n0 = nb * 32;
    const int Hv = UPS ? 2 * p.Hin : p.Hin, Wv = UPS ? 2 * p.Win : p.Win;
    const int bw = Wv >> 5, bh = Hv >> 4;
    const int img = tb / (bw * bh), brem = tb - img * (bw * bh);
    const int y0 = (brem / bw) * 16, x0 = (brem - (brem / bw) * bw) * 32;
    const int nkt = p.Cin >> 3;
    // BLK: the input is the channel-blocked copy k_gn_apply_blk wrote, [Cin/8][pixel][8]: a k-tile's 8 channels of a pixel are 32 bytes, the
    // pixels of an image row follow each other, a k-tile is a plane of kstep bytes.  Otherwise NHWC: 32 bytes at a pitch of Cin*4.
    static_assert(!(UPS && BLK), "the upsampling convolution reads a raw NHWC tensor");
    const unsigned pitch4 = BLK ? 32u : (unsigned)p.in_pitch * 4u;
    const int kstep = BLK ? (int)(p.M * 32) : 32;
    const int kt0 = blockIdx.z * p.kt_per;                        // split-K over input channels: this slab's k-tiles
    const int ntiles = min(nkt, kt0 + p.kt_per) - kt0;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.in, (short)0, (int)((long)p.N * p.Hin * p.Win * p.in_pitch * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.w_wino, (short)0, (int)((long)p.n_nblocks * nkt * U_F * 4), 0x00020000);

    f32x16 acc[9];
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // everything inside the K walk is compiled once per wave: its frequency block (FR, FC) and its share of the patch DMAs
    auto run = [&](auto wc) {
        constexpr int W = decltype(wc)::value, FR = W >> 1, FC = W & 1, NPW = (NP - W + 3) / 4;
        unsigned pv[NPW];
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int c = (W + 4 * j) * 64 + lane;                    // patch chunk: lanes 2i, 2i+1 fetch the two halves of one pixel
            int row, col, h;
            if (BLK) {   // eight lanes = the four pixels 4g..4g+3 of a row = 128 contiguous bytes, order inside the group swizzled
                row = c / 72;
                const int g = (c - row * 72) >> 3, kk = (c & 7) ^ ((((g >> 1) & 1) << 2) | ((row >> 2) & 3));
                col = 4 * g + (kk >> 1); h = kk & 1;
            } else {
                const int pixp = c >> 1, q = pixp % PRW;
                row = pixp / PRW; h = (c & 1) ^ ((row >> 2) & 1);
                col = (q % 9) * 4 + q / 9;
            }
            const int y = y0 - 1 + row, x = x0 - 1 + col;
            const bool ok = c < P_REAL && col < 34 && y >= 0 && y < Hv && x >= 0 && x < Wv;
            const int ys = UPS ? y >> 1 : y, xs = UPS ? x >> 1 : x;
            pv[j] = ok ? (unsigned)((img * p.Hin + ys) * p.Win + xs) * pitch4 + h * 16 : OOB;
        }
        int soffA = kt0 * kstep;
        int soffU = ((nb * nkt + kt0) * 36 + W * 9) * 1024;           // this wave's first slice of k-tile kt0
        const unsigned uvp = (unsigned)lane * 16u;
        const int T = lane & 31, ty = T >> 3, tx = T & 7;
        // window row rr of this lane's tile is patch row 4*ty + FR + rr; its pixels' halves are swapped when bit 2 of the row is set
        const float *pread = lds + U_F + (((4 * ty + FR) * PRW + tx) * 2 + (half ^ (ty & 1))) * 4;   // + stage*P_F + ((rr*PRW + coff(c))*2 +- 1)*4
        // BLK: byte offset of (row, group g = tx + (c>>2), slot ((c&3)*2 + half) ^ swizzle(row, g)); the swizzle has four per-lane variants
        int Av[2][2];
#pragma unroll
        for (int cq = 0; cq < 2; ++cq)
#pragma unroll
            for (int rq = 0; rq < 2; ++rq)
                Av[cq][rq] = ((4 * ty + FR) * 9 + tx) * 128 + ((half ^ (((((tx + cq) >> 1) & 1) << 2) | ((ty + rq) & 3))) << 4);
        const float *uread = lds + W * 9 * 256 + lane * 4;                                // + f*256

        // one piece (64 chunks) of the patch of the k-tile at soffA; the last one is ragged (P_REAL is not a multiple of 64): its tail
        // lanes are switched off, so that a stage is exactly P_REAL chunks and two workgroups fit a CU's 160 KB
        auto issue_piece = [&](int stage, auto jc) {
            constexpr int j = decltype(jc)::value;
            if ((W + 4 * j + 1) * 64 <= P_REAL || lane < P_REAL - (W + 4 * j) * 64)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsA, (__attribute__((address_space(3))) void *)(lds + U_F + stage * P_F + (W + 4 * j) * 256), 16, pv[j], soffA, 0, 0);
        };
        auto issue_u = [&](int f) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void *)(lds + (W * 9 + f) * 256), 16, uvp,
                                                     soffU + f * 1024, 0, 0);
        };
        f32x4 V[9], ub[2];
        // columns in the order the horizontal sums want them (each partial is one fma chain), so that only three partials per
        // frequency row are live next to the column being transformed: (0,+a,-a) needs 4,2,0,3,1; (+b,-b,inf) needs 3,1,4,2,0
        f32x4 x[5];
        auto load_col = [&](int stage, int k) {
            constexpr int ord0[5] = {4, 2, 0, 3, 1}, ord1[5] = {3, 1, 4, 2, 0};
            const int c = FC + (FC == 0 ? ord0[k] : ord1[k]), coff = ((c & 3) * 9 + (c >> 2)) * 8;
            if (BLK) {
                const char *sb = reinterpret_cast<const char *>(lds + U_F + stage * P_F);
#pragma unroll
                for (int rr = 0; rr < 5; ++rr)
                    x[rr] = *reinterpret_cast<const f32x4 *>(sb + ((Av[c >> 2][(FR + rr) >> 2] ^ ((c & 3) << 5)) + (rr * 9 + (c >> 2)) * 128));
                return;
            }
            const float *pb = pread + stage * P_F + coff;
            const int flip = (half ^ (ty & 1)) ? -4 : 4;              // rows 4, 5 of the window: the other half-slot of the pixel
#pragma unroll
            for (int rr = 0; rr < 5; ++rr) x[rr] = *reinterpret_cast<const f32x4 *>(pb + rr * PRW * 8 + (((FR + rr) >> 2) & 1) * flip);
        };
        auto transform = [&](int stage) {   // the first column is already on its way (load_col(stage, 0))
            f32x4 P[3][3];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                f32x4 o[3];
                w4_fwd<FR>(x, o);
                // (asm pins: the instruction selector otherwise issues all 25 reads first and keeps them live)
                asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]) : : "memory");
                if (k < 4) load_col(stage, k + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ii = 0; ii < 3; ++ii) {
                    const f32x4 y = o[ii];
                    if constexpr (FC == 0) {
                        if (k == 0) P[ii][0] = y;
                        else if (k == 1) { P[ii][1] = fma4(-W4_B * W4_B, y, P[ii][0]); P[ii][0] = fma4(W4_C2, y, P[ii][0]); }
                        else if (k == 2) P[ii][0] = fma4(W4_C0, y, P[ii][0]);
                        else if (k == 3) P[ii][2] = y;
                        else P[ii][2] = fma4(-W4_B * W4_B, y, P[ii][2]);
                    } else {
                        if (k == 0) P[ii][0] = y;
                        else if (k == 1) P[ii][0] = fma4(-W4_A * W4_A, y, P[ii][0]);
                        else if (k == 2) P[ii][2] = y;
                        else if (k == 3) { P[ii][2] = fma4(W4_C2, y, P[ii][2]); P[ii][1] = y; }
                        else { P[ii][1] = fma4(-W4_A * W4_A, y, P[ii][1]); P[ii][2] = fma4(W4_C0, y, P[ii][2]); }
                    }
                }
                auto pin = [&](int j) { asm volatile("" : "+v"(P[0][j]), "+v"(P[1][j]), "+v"(P[2][j]) : : "memory"); };
                if (k < 4) {
                    pin(0);
                    if (FC == 0 ? k >= 1 : k >= 3) pin(1);
                    if (FC == 0 ? k >= 3 : k >= 2) pin(2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ii = 0; ii < 3; ++ii) {
                if constexpr (FC == 0) {
                    V[ii * 3 + 0] = P[ii][0];
                    V[ii * 3 + 1] = fma4(W4_A, P[ii][2], P[ii][1]);
                    V[ii * 3 + 2] = fma4(-W4_A, P[ii][2], P[ii][1]);
                } else {
                    V[ii * 3 + 0] = fma4(W4_B, P[ii][1], P[ii][0]);
                    V[ii * 3 + 1] = fma4(-W4_B, P[ii][1], P[ii][0]);
                    V[ii * 3 + 2] = P[ii][2];
                }
            }
            // the transform is finished HERE: without a use in this block the compiler sinks the arithmetic behind the barrier
            // and keeps all 25 window reads live instead (spills)
            asm volatile("" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7]), "+v"(V[8]));
        };
        // Per-wave DMA queue of a k-tile, issued behind the MFMAs of its units:  U(t+1,0) P0(t+2) U(t+1,1) P1(t+2) ... U(t+1,8)  (NPW patch
        // pieces behind the first NPW units - one piece per unit instead of a burst of 21 per workgroup behind the barrier).  Every wait
        // is an exact count of the younger instructions of that fixed sequence:
        //   slot f+1, read in unit f (issued one k-tile ago): 6 + NPW younger while f < NPW, 7 + NPW after; the first read: 8 + NPW
        //   the patch of k-tile t+1 at the barrier behind transform(t): the 9 - NPW weight slots issued behind its last piece
        // MODE 0: steady state; 1: the k-tile before the last (no patch pieces any more: 7 + max(0, NPW - f - 1), first 8 + NPW);
        // 2: the last one (nothing is issued, the queue drains).
        auto piece = [&](int stage, int j) {
            if (j == 0) issue_piece(stage, std::integral_constant<int, 0>{});
            else if (j == 1) issue_piece(stage, std::integral_constant<int, 1>{});
            else if (j == 2) issue_piece(stage, std::integral_constant<int, 2>{});
            else if (j == 3) issue_piece(stage, std::integral_constant<int, 3>{});
            else if (j == 4) issue_piece(stage, std::integral_constant<int, 4>{});
            else if (NPW > 5 && j == 5) issue_piece(stage, std::integral_constant<int, (NPW > 5 ? 5 : 0)>{});
        };
        if (ntiles > 0) {
#pragma unroll
            for (int j = 0; j < NPW; ++j) piece(0, j);
            soffA += kstep;
#pragma unroll
            for (int f = 0; f < 9; ++f) {
                issue_u(f);
                if (ntiles > 1 && f < NPW) piece(1, f);
            }
            soffA += kstep;
            soffU += 36 * 1024;
            if (ntiles > 1) wait_vmcnt<9 + NPW>(); else wait_vmcnt<9>();       // patch 0 has landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            load_col(0, 0);
        }
        // STEADY: k-tiles t with t + 2 < ntiles (everything known at compile time, no branches in the loop); the last two k-tiles of a
        // workgroup run the same code with the conditions evaluated at run time
        auto body = [&](auto sc, auto stc, int t) {
            constexpr int S = decltype(sc)::value;
            constexpr bool STEADY = decltype(stc)::value;
            const bool more = STEADY || t + 1 < ntiles, more2 = STEADY || t + 2 < ntiles;
            transform(S);
            if (more) {
                // every wave has read patch t; patch t+1 is published
                wait_vmcnt<9 - NPW>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (more) wait_vmcnt<8 + NPW>(); else wait_vmcnt<0>();
            ub[0] = *reinterpret_cast<const f32x4 *>(uread);
#pragma unroll
            for (int f = 0; f < 9; ++f) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[f][0], ub[f & 1][0], acc[f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (f < 8) {
                    if (more2) { if (f < NPW) wait_vmcnt<6 + NPW>(); else wait_vmcnt<7 + NPW>(); }
                    else if (more) {
                        const int k = NPW - f - 1;                    // patch pieces of k-tile t+1 that sit behind slot f+1 in the queue
                        if (k >= 5) wait_vmcnt<12>(); else if (k == 4) wait_vmcnt<11>(); else if (k == 3) wait_vmcnt<10>();
                        else if (k == 2) wait_vmcnt<9>(); else if (k == 1) wait_vmcnt<8>(); else wait_vmcnt<7>();
                    }
                    ub[(f + 1) & 1] = *reinterpret_cast<const f32x4 *>(uread + (f + 1) * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 1; s < 4; ++s) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[f][s], ub[f & 1][s], acc[f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) issue_u(f);                                 // the slot just consumed refills for k-tile t+1
                if (more2 && f < NPW) piece(S, f);                    // a piece of the patch of k-tile t+2 into the stage just released
                if (f == 6 && more) load_col(S ^ 1, 0);               // first window column of k-tile t+1 (published at the barrier above)
            }
            if (more) soffU += 36 * 1024;
            if (more2) soffA += kstep;
        };
        if (ntiles > 0) {
            using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
            int t = 0;
            for (; t + 3 < ntiles; t += 2) { body(S0{}, std::true_type{}, t); body(S1{}, std::true_type{}, t + 1); }
            for (; t < ntiles; t += 2) {
                body(S0{}, std::false_type{}, t);
                if (t + 1 < ntiles) body(S1{}, std::false_type{}, t + 1);
            }
        }
    };
    switch (wave) {
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        default: run(std::integral_constant<int, 3>{}); break;
    }

    // the 36 frequencies of a (tile, channel) meet in LDS: two rounds of 16 tiles; [freq][tile][cout].  A thread then finishes one tile
    // for two neighbouring output channels (ds_read_b64, 8-byte stores: 16 lanes cover the 128 bytes of a pixel's 32 channels)
    const int fbase = (3 * (wave >> 1)) * 6 + 3 * (wave & 1);
    const int mloc = tid >> 4, np2 = (tid & 15) * 2, n = n0 + np2;
    f32x2 bs = {0.f, 0.f};
    if (p.bias && !p.partial) {   // (ragged Cout - only the NCHW output convolution: channels past Cout are computed from zero weights and not stored)
        if (n + 1 < p.Cout) bs = *reinterpret_cast<const f32x2 *>(p.bias + n);
        else if (n < p.Cout) bs[0] = p.bias[n];
    }
    const long hw = (long)Hv * Wv;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        __syncthreads();
#pragma unroll
        for (int f = 0; f < 9; ++f) {
            const int F = fbase + (f / 3) * 6 + (f % 3);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = 8 * q + rr, ml = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
                lds[(F * 16 + ml) * 32 + (lane & 31)] = acc[f][r];
            }
        }
        __syncthreads();
        const float *zz = lds + mloc * 32 + np2;
        f32x2 z[4][6];                                                // rows of A^T applied: z[p][j]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x2 col[4];
            w4_out(*reinterpret_cast<const f32x2 *>(zz + (0 * 6 + j) * 512), *reinterpret_cast<const f32x2 *>(zz + (1 * 6 + j) * 512),
                   *reinterpret_cast<const f32x2 *>(zz + (2 * 6 + j) * 512), *reinterpret_cast<const f32x2 *>(zz + (3 * 6 + j) * 512),
                   *reinterpret_cast<const f32x2 *>(zz + (4 * 6 + j) * 512), *reinterpret_cast<const f32x2 *>(zz + (5 * 6 + j) * 512), col);
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) z[pr][j] = col[pr];
        }
        const int Tg = q * 16 + mloc, oy = y0 + 4 * (Tg >> 3), ox = x0 + 4 * (Tg & 7);
        const long m0 = ((long)img * Hv + oy) * Wv + ox;              // pixel (pr, qc) of the tile: m0 + pr*Wv + qc
        f32x2 v[16];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            f32x2 row[4];
            w4_out(z[pr][0], z[pr][1], z[pr][2], z[pr][3], z[pr][4], z[pr][5], row);
#pragma unroll
            for (int qc = 0; qc < 4; ++qc) v[pr * 4 + qc] = row[qc] + bs;
        }
        if (p.partial) {   // split-K: the output transform is linear, so slabs are summed in the output domain by k_splitk_finish
            float *dst = p.partial + (long)blockIdx.z * p.M * p.Cout + m0 * p.Cout + n;
#pragma unroll
            for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x2 *>(dst + (long)((k >> 2) * Wv + (k & 3)) * p.Cout) = v[k];
            continue;
        }
        if (p.res) {
            const float *rp = p.res + m0 * p.res_pitch + n;
            f32x2 rr[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) rr[k] = *reinterpret_cast<const f32x2 *>(rp + (long)((k >> 2) * Wv + (k & 3)) * p.res_pitch);
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] += rr[k];
        }
        f32x2 v2[16];
        if (p.out2) {
            const float *rp = p.res2 + m0 * p.res2_pitch + n;
#pragma unroll
            for (int k = 0; k < 16; ++k) v2[k] = *reinterpret_cast<const f32x2 *>(rp + (long)((k >> 2) * Wv + (k & 3)) * p.res2_pitch);
#pragma unroll
            for (int k = 0; k < 16; ++k) v2[k] += v[k];
        }
        // GroupNorm statistics: the wave's four tiles of the round = 64 pixels of one image
        auto stats = [&](float *red, const f32x2(&vv)[16]) {   // (deposited in the statistics corner of LDS; the workgroup adds one pair per channel below)
            f32x2 sm = {0.f, 0.f}, sq = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) { sm += vv[k]; sq += vv[k] * vv[k]; }
            f32x4 r = {sm[0], sq[0], sm[1], sq[1]};
#pragma unroll
            for (int i = 0; i < 4; ++i) { r[i] += __shfl_xor(r[i], 16); r[i] += __shfl_xor(r[i], 32); }
            if (lane < 16) { wg_stat_put<32>(red, q * 4 + wave, np2, r[0], r[1]); wg_stat_put<32>(red, q * 4 + wave, np2 + 1, r[2], r[3]); }
        };
        // (floats 18432.. of the LDS image lie behind the 36 x 16 x 32 exchange buffer: 8 (round, wave) parts x 32 channels x 2, twice)
        if (p.st1) stats(lds + 18432, v);
        if (p.st2) stats(lds + 18432 + 512, v2);
        if (p.out_nchw) {
            float *op = p.out + ((long)img * p.Cout + n) * hw + (m0 - (long)img * hw);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (n < p.Cout) op[(k >> 2) * Wv + (k & 3)] = v[k][0];
                if (n + 1 < p.Cout) op[hw + (k >> 2) * Wv + (k & 3)] = v[k][1];
            }
        } else {
            float *op = p.out + m0 * p.out_pitch + n;
#pragma unroll
            for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x2 *>(op + (long)((k >> 2) * Wv + (k & 3)) * p.out_pitch) = v[k];
        }
        if (p.out2) {
            float *op = p.out2 + m0 * p.out2_pitch + n;
#pragma unroll
            for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x2 *>(op + (long)((k >> 2) * Wv + (k & 3)) * p.out2_pitch) = v2[k];
        }
    }
    if (p.st1 || p.st2) {   // one pair per channel of the workgroup's 512 pixels (of one image) into the fixed-point totals
        __syncthreads();
        if (p.st1) wg_group_flush<32, 8>(lds + 18432, reinterpret_cast<unsigned long long *>(lds + 19456), p.st1, p.N, img, p.Cout, n0, p.st1_c0, p.st1_cg, hw, tid);
        if (p.st2) wg_group_flush<32, 8>(lds + 18432 + 512, reinterpret_cast<unsigned long long *>(lds + 19456), p.st2, p.N, img, p.Cout, n0, p.st2_c0, p.st2_cg, hw, tid);
    }
#endif
}

// weights -> U = G g G^T (6x6) per (cout, cin) for k_conv_wino4: [cout/32][cin/8][wave 4][f 9][half][32][4]; wave (fr, fc), f = 3*ii + jj:
// frequency (3*fr + ii, 3*fc + jj) in the order (0, +a, -a, +b, -b, inf)
// (thread per (output channel, input channel), like k_pack_conv_wino: a block = 32 output channels x one k-tile, 36 frequencies)
__global__ __launch_bounds__(256) void k_pack_conv_wino4(const float *__restrict__ w, int Cout, int Cin, int Cin_pad, float *__restrict__ dst, int tf) {
    const int nkt = Cin_pad >> 3;
    const double a = W4_A, b = W4_B, n0 = a * a * b * b, na = 2 * a * a * (a * a - b * b), nb_ = 2 * b * b * (b * b - a * a);
    const double G[6][3] = {{1 / n0, 0, 0}, {1 / na, a / na, a * a / na}, {1 / na, -a / na, a * a / na},
                            {1 / nb_, b / nb_, b * b / nb_}, {1 / nb_, -b / nb_, b * b / nb_}, {0, 0, 1}};
    const int t = threadIdx.x, s = t & 3, nn = (t >> 2) & 31, hf = t >> 7;
    const long nblk = (long)((Cout + 31) >> 5) * nkt;          // (output channels past Cout: zero rows)
    for (long bi = blockIdx.x; bi < nblk; bi += gridDim.x) {
        const int kt = (int)(bi % nkt), nb = (int)(bi / nkt);
        const int co = nb * 32 + nn, ci = kt * 8 + hf * 4 + s;
        const bool ok = ci < Cin && co < Cout;
        double g[9];
        if (ok) {
            const float *q = tf ? w + ((long)ci * Cout + co) * 9 : w + ((long)co * Cin + ci) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) g[k] = (double)q[tf ? 8 - k : k];
        }
        float *o = dst + (((long)nb * nkt + kt) * 36 * 2 + hf) * 128 + nn * 4 + s;
#pragma unroll
        for (int wf = 0; wf < 36; ++wf) {
            const int wv = wf / 9, f = wf - wv * 9;
            const int i = 3 * (wv >> 1) + f / 3, j = 3 * (wv & 1) + f % 3;
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int vv = 0; vv < 3; ++vv) acc += G[i][u] * g[u * 3 + vv] * G[j][vv];
            o[(long)wf * 256] = ok ? (float)acc : 0.f;
        }
    }
}

// k_conv_bf3: k_conv_dma with the fp32 products emulated on the bf16 matrix pipe (opt-in, see hl_unet_set_conv_mode).
// NPL = 3: the emulation above.  NPL = 1: the 16-bit training arithmetic (hl_unet_set_conv_mode HL_CONV_BF16, what the reference's
// autocast selects, train_util.py:214): every activation is rounded to bf16 (nearest-even) and multiplied with the weight's two leading
// bf16 planes (16 significand bits - the packed planes are already there), fp32 accumulation: two bf16 MFMAs per k-tile instead of six.
template <int WM, bool UPS, int NPL = 3>
__global__ __launch_bounds__(WM * 64, (WM == 8) ? 4 : 3) void k_conv_bf3(const ConvK p) {
#if __HIP_DEVICE_COMPILE__   // device pass only: the host pass of this clang drops the launch stub when it parses the body
    constexpr int NS = 3;
    constexpr int BM = WM * 32, BN = 96, A_F = BM * 16, B_F = 6 * BN * 4, STAGE_F = A_F + B_F;
    constexpr int NBJ = (9 + WM - 1) / WM;   // B instructions per wave at most (9 in total, instruction b = wave + WM*j)
    constexpr unsigned OOB = 0x80000000u;    // buffer offset past num_records (< 2 GiB): the load returns zeros
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = p.n_mtiles * p.n_nblocks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int wi = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
    const int mt_idx = wi / p.n_nblocks;
    const long m0 = (long)mt_idx * BM;
    const int n0 = (wi - mt_idx * p.n_nblocks) * BN;
    const int pad = p.ks >> 1;
    const int ncc = p.Cin >> 4;
    const int hw_out = p.Hout * p.Wout;
    const unsigned pitch4 = (unsigned)p.in_pitch * 4u;

    // Both operands are fetched through buffer descriptors: address = base + per-lane offset (VGPR) + a wave-uniform
    // offset (SGPR), so walking K costs no vector ALU at all, and padding / ragged rows are lanes whose offset is
    // out of range (hardware returns zeros).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.in, (short)0, (int)((long)p.N * p.Hin * p.Win * p.in_pitch * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.w_bf3, (short)0, (int)((long)p.wrows * p.Ktot * 6), 0x00020000);

    // DMA instruction i of a tile fills rows 16i..16i+15: A rows by instructions wave + WM*j (j = 0,1), the 96 B rows
    // by instructions BM/16 + b with b = wave + WM*j < 6.
    // Lane L of an instruction writes physical quarter L&3 of row 16i + (L>>2).
    const int lrow = lane >> 2, pq = lane & 3;
    int iy0[2], ix0[2];    // top-left input coordinate of the 3x3 window (in the x2 grid when UPS); huge negative = no row
    unsigned nb[2];        // byte offset of image n, plus this lane's quarter
    unsigned a_cur[2];     // byte offset of this lane's 16 bytes for the current tap (or OOB)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave + WM * j) * 16 + lrow;
        const int ql = pq ^ ((row >> 2) & 3);
        const long P = m0 + row;
        const bool in = P < p.M;
        const long Pc = in ? P : 0;
        const int n = (int)(Pc / hw_out);
        const int rem = (int)(Pc - (long)n * hw_out);
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        iy0[j] = in ? oy * p.stride - pad : -(1 << 20);
        ix0[j] = ox * p.stride - pad;
        nb[j] = (unsigned)n * (unsigned)(p.Hin * p.Win) * pitch4 + ql * 16;
    }
    auto set_tap = [&](int tap) {
        const int ky = (p.ks == 3) ? tap / 3 : 0, kx = (p.ks == 3) ? tap - ky * 3 : 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int iy = iy0[j] + ky, ix = ix0[j] + kx;
            const int Hv = UPS ? 2 * p.Hin : p.Hin, Wv = UPS ? 2 * p.Win : p.Win;
            const bool ok = iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
            const int sy = UPS ? iy >> 1 : iy, sx = UPS ? ix >> 1 : ix;
            a_cur[j] = ok ? nb[j] + (unsigned)(sy * p.Win + sx) * pitch4 : OOB;
        }
    };
    const int nk_all = ncc * p.taps;
    const int kt0 = blockIdx.z * p.kt_per;
    const int nk = min(nk_all, kt0 + p.kt_per);
    // B stage: 576 chunks of 16 bytes, chunk (part, n) at index part*96 + n, part = plane*2 + k-half (8 bf16 each):
    // the 8 lanes a ds_read_b128 serves per cycle read 8 consecutive chunks.  DMA instruction b (0..8), lane L fills
    // chunk 64b + L from the packed weights [row][k-tile][part][8 bf16].
    unsigned b_voff[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
        const int c = (wave + WM * j) * 64 + lane;
        const int part = c / BN, nn = c - part * BN;
        const int gn = n0 + nn;
        b_voff[j] = (c < 6 * BN && gn < p.wrows) ? (unsigned)gn * (unsigned)p.Ktot * 6u + part * 16 : OOB;
    }
    int n_b = 0;                                               // B instructions of this wave
#pragma unroll
    for (int j = 0; j < NBJ; ++j) n_b += (wave + WM * j < 9) ? 1 : 0;
    const int n_w = 2 + n_b;                                   // DMA instructions of this wave per tile

    int tap_i, cc_i;                                           // issue cursor
    kt_decode(kt0, ncc, p.taps, cc_i, tap_i);
    int cg_i = cc_i / KG, gend_i = min(ncc, (cg_i + 1) * KG);
    int soffB = kt0 * 96;
    set_tap(tap_i);
    auto issue = [&](int stage) {
        float *dst = lds + stage * STAGE_F + wave * 256;       // + WM*256 floats per j (WM instructions further)
        const int soffA = cc_i * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(dst + j * (WM * 256)), 16,
                                                     a_cur[j], soffA, 0, 0);
        float *dstb = lds + stage * STAGE_F + A_F + wave * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)dstb, 16, b_voff[0], soffB, 0, 0);
        if (wave + WM < 9)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(dstb + WM * 256), 16,
                                                     b_voff[1], soffB, 0, 0);
        if (NBJ > 2 && wave + 2 * WM < 9)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(dstb + 2 * WM * 256), 16,
                                                     b_voff[NBJ > 2 ? 2 : 0], soffB, 0, 0);
        soffB += 96;
        if (++cc_i == gend_i) {
            if (++tap_i == p.taps) { tap_i = 0; ++cg_i; gend_i = min(ncc, (cg_i + 1) * KG); }
            cc_i = cg_i * KG;
            set_tap(tap_i);
        }
    };
    auto wait_younger = [&]() {   // all but this wave's DMAs of the youngest tile have landed
        if (n_w == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (n_w == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    };

    // fragment read offsets (floats) inside a stage
    const int ra = wave * 32 + (lane & 31);
    const int sa = (ra >> 2) & 3;
    const int a_off0 = ra * 16 + (((2 * half) ^ sa) << 2), a_off1 = ra * 16 + (((2 * half + 1) ^ sa) << 2);
    const int b_off = A_F + (half * BN + (lane & 31)) * 4;     // + (plane*2*96 + j*32) * 4 floats

    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // One v_mfma_f32_32x32x16_bf16 contracts the whole 16-channel k-tile (lane half h holds channels 8h..8h+7, exactly
    // the two 16-byte quarters this lane reads of its A row).  fp32 x fp32 is emulated by splitting both operands into
    // three bf16 planes (x = hi + mid + lo exactly: 3 x 8 significand bits, by truncation) and accumulating the six
    // products whose weight is >= 2^-16 of the leading one in the fp32 accumulator:
    //     a*b ~= ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)        (dropped terms <= 3 * 2^-24 |a*b|)
    // Weights are split once at pack time; the activation fragment is split here, after the LDS read.
    const int ntiles = nk - kt0;
    auto split3 = [&](const f32x4 x0, const f32x4 x1, bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
        if constexpr (NPL == 1) {   // round to nearest even: u + 0x7fff + lsb, upper half (no NaN / overflow care needed beyond what fp32 gives)
            u32x4 ph;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned a = __float_as_uint(2 * i < 4 ? x0[2 * i] : x1[2 * i - 4]), b = __float_as_uint(2 * i + 1 < 4 ? x0[2 * i + 1] : x1[2 * i + 1 - 4]);
                const unsigned ra = a + 0x7fffu + ((a >> 16) & 1u), rb = b + 0x7fffu + ((b >> 16) & 1u);
                ph[i] = __builtin_amdgcn_perm(rb, ra, 0x07060302);
            }
            hi = __builtin_bit_cast(bf16x8, ph);
            return;
        }
        unsigned u[8], m[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = i < 4 ? x0[i] : x1[i - 4];
            u[i] = __float_as_uint(x);
            const float r1 = x - __uint_as_float(u[i] & 0xffff0000u);      // exact
            m[i] = __float_as_uint(r1);
            const float r2 = r1 - __uint_as_float(m[i] & 0xffff0000u);     // exact, <= 8 significant bits
            l[i] = __float_as_uint(r2);
        }
        u32x4 ph, pm, pl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // two truncated bf16 per dword: element 2i low, 2i+1 high
            ph[i] = __builtin_amdgcn_perm(u[2 * i + 1], u[2 * i], 0x07060302);
            pm[i] = __builtin_amdgcn_perm(m[2 * i + 1], m[2 * i], 0x07060302);
            pl[i] = __builtin_amdgcn_perm(l[2 * i + 1], l[2 * i], 0x07060302);
        }
        hi = __builtin_bit_cast(bf16x8, ph);
        mid = __builtin_bit_cast(bf16x8, pm);
        lo = __builtin_bit_cast(bf16x8, pl);
    };
    // Pipeline in units of (k-tile t, 32-column block j) = 6 MFMAs: the weight planes of the next unit are read from LDS
    // while the current unit multiplies; the barrier for tile t+1 sits in front of unit (t, 2), whose operands are
    // already in registers, and the next activation fragment is read and split behind those MFMAs.
    //   unit (t,0): read B(t,1) | 6 mfma     unit (t,1): read B(t,2) | 6 mfma
    //   unit (t,2): wait DMA(t+1), barrier | issue DMA(t+3) | read A(t+1), B(t+1,0) | 6 mfma | split A(t+1)
    auto readB = [&](const float *base, int j, bf16x8 &h, bf16x8 &m, bf16x8 &l) {
        h = *reinterpret_cast<const bf16x8 *>(base + b_off + (0 * BN + j * 32) * 4);
        m = *reinterpret_cast<const bf16x8 *>(base + b_off + (2 * BN + j * 32) * 4);
        if constexpr (NPL == 3) l = *reinterpret_cast<const bf16x8 *>(base + b_off + (4 * BN + j * 32) * 4);
    };
    auto mma6 = [&](f32x16 &c, const bf16x8 ah, const bf16x8 am, const bf16x8 al, const bf16x8 bh, const bf16x8 bm, const bf16x8 bl) {
        if constexpr (NPL == 3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);   // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    };
    bf16x8 ah, am, al, b0h, b0m, b0l, b1h, b1m, b1l;
    if (ntiles > 0) {
        issue(0);
        if (ntiles > 1) { issue(1); wait_younger(); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ntiles > 2) issue(2);
        const f32x4 x0 = *reinterpret_cast<const f32x4 *>(lds + a_off0);
        const f32x4 x1 = *reinterpret_cast<const f32x4 *>(lds + a_off1);
        readB(lds, 0, b0h, b0m, b0l);
        split3(x0, x1, ah, am, al);
    }
    auto body = [&](auto uc, int t) {
        constexpr int U = decltype(uc)::value, UN = (U + 1) % NS;
        const float *base = lds + U * STAGE_F, *nbase = lds + UN * STAGE_F;
        readB(base, 1, b1h, b1m, b1l);
        __builtin_amdgcn_sched_barrier(0);
        mma6(acc[0], ah, am, al, b0h, b0m, b0l);
        __builtin_amdgcn_sched_barrier(0);
        readB(base, 2, b0h, b0m, b0l);
        __builtin_amdgcn_sched_barrier(0);
        mma6(acc[1], ah, am, al, b1h, b1m, b1l);
        __builtin_amdgcn_sched_barrier(0);
        const bool more = t + 1 < ntiles;
        f32x4 x0, x1;
        if (more) {
            if (t + 2 < ntiles) wait_younger(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every wave is done reading tile t before its stage refills
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // unit (t,2): its weight planes are in b0*, so first park them, then start the reads of tile t+1
        const bf16x8 ch = b0h, cm = b0m, cl = b0l;
        f32x16 c2 = acc[2];
        if constexpr (NPL == 3) c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ch, c2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            if (t + NS < ntiles) issue(U);
            x0 = *reinterpret_cast<const f32x4 *>(nbase + a_off0);
            x1 = *reinterpret_cast<const f32x4 *>(nbase + a_off1);
            readB(nbase, 0, b0h, b0m, b0l);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NPL == 3) {
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, cm, c2, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cl, c2, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, ch, c2, 0, 0, 0);
        }
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cm, c2, 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ch, c2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) split3(x0, x1, ah, am, al);
    };
    for (int t = 0; t < ntiles; t += NS) {
        body(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) body(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < ntiles) body(std::integral_constant<int, 2>{}, t + 2);
    }

    // epilogue (same contract as k_conv)
    if (p.partial) {
        float *dst = p.partial + (long)blockIdx.z * p.M * p.Cout;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = n0 + j * 32 + (lane & 31);
            if (n >= p.Cout) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < p.M) dst[m * p.Cout + n] = acc[j][r];
            }
        }
        return;
    }
    // per 32-column block: all loads (residual, second residual) are issued before any store, so they overlap
    // instead of serialising behind the stores (res may alias out)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int n = n0 + j * 32 + (lane & 31);
        if (n >= p.Cout) continue;
        const float bs = p.bias ? p.bias[n] : 0.f;
        const long mb = m0 + wave * 32 + 4 * half;
        float v[16], v2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[j][r] + bs;
        if (p.res) {
            float rr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = mb + (r & 3) + 8 * (r >> 2);
                rr[r] = m < p.M ? p.res[m * p.res_pitch + n] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += rr[r];
        }
        if (p.out2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = mb + (r & 3) + 8 * (r >> 2);
                v2[r] = m < p.M ? p.res2[m * p.res2_pitch + n] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v2[r] += v[r];
        }
        if (p.out_nchw) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = mb + (r & 3) + 8 * (r >> 2);
                const long img = m / hw_out, rem = m - img * hw_out;
                if (m < p.M) p.out[(img * p.Cout + n) * hw_out + rem] = v[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = mb + (r & 3) + 8 * (r >> 2);
                if (m < p.M) p.out[m * p.out_pitch + n] = v[r];
            }
        }
        if (p.out2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = mb + (r & 3) + 8 * (r >> 2);
                if (m < p.M) p.out2[m * p.out2_pitch + n] = v2[r];
            }
        }
    }
#endif
}

// y = x*A[n,c] + B[n,c] (and SiLU): the GroupNorm-apply pre-pass for k_conv_dma.  x has a channel pitch, y is dense.
__global__ void k_gn_apply(const float *__restrict__ x, long pitch, long pixels_per_img, long npix, int C,
                           const float *__restrict__ cA, const float *__restrict__ cB, int act, float *__restrict__ y) {
    const int cq = C >> 2;
    const long n4 = npix * cq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cq;
        const int c = (int)(i - pix * cq) * 4;
        const long n = pix / pixels_per_img;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + pix * pitch + c);
        const f32x4 a = *reinterpret_cast<const f32x4 *>(cA + n * C + c);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(cB + n * C + c);
        f32x4 o = v * a + b;
        if (act) { o[0] = silu_f(o[0]); o[1] = silu_f(o[1]); o[2] = silu_f(o[2]); o[3] = silu_f(o[3]); }
        *reinterpret_cast<f32x4 *>(y + pix * C + c) = o;
    }
}

// the two fp16 planes of four values (fp16x2 products, split_h2 of hl_conv_h16.hip: h0 = the nearest fp16, h1 = the nearest fp16 of the residual): image [plane][pixel][C]
__device__ __forceinline__ void store_h2_planes(unsigned short *y, long at, long plane, const f32x4 o) {
    unsigned q0[2], q1[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float a = o[2 * k], b = o[2 * k + 1];
        hl_split2_rne(a, b, q0[k], q1[k]);
    }
    *reinterpret_cast<uint2 *>(y + at) = uint2{q0[0], q0[1]};
    *reinterpret_cast<uint2 *>(y + plane + at) = uint2{q1[0], q1[1]};
}

// the same pass writing 16-bit values (fp16 / bf16, nearest even; f16 = 2: the two fp16x2 planes): the activation image of the k_conv_h16 /
// k_conv1_h16 layers behind a GroupNorm - half the bytes written here and read there, no rounding in the convolution's staging
__global__ void k_gn_apply_h16(const float *__restrict__ x, long pitch, long pixels_per_img, long npix, int C, const float *__restrict__ cA,
                               const float *__restrict__ cB, int act, unsigned short *__restrict__ y, int f16) {
    const int cq = C >> 2;
    const long n4 = npix * cq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cq;
        const int c = (int)(i - pix * cq) * 4;
        const long n = pix / pixels_per_img;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + pix * pitch + c);
        const f32x4 a = *reinterpret_cast<const f32x4 *>(cA + n * C + c);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(cB + n * C + c);
        f32x4 o = v * a + b;
        if (act) { o[0] = silu_f(o[0]); o[1] = silu_f(o[1]); o[2] = silu_f(o[2]); o[3] = silu_f(o[3]); }
        if (f16 == 2) { store_h2_planes(y, pix * C + c, npix * C, o); continue; }
        unsigned short h[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (f16) {
                const _Float16 hh = (_Float16)o[k];
                h[k] = __builtin_bit_cast(unsigned short, hh);
            } else {
                const unsigned u = __float_as_uint(o[k]);
                h[k] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            }
        }
        *reinterpret_cast<uint2 *>(y + pix * C + c) = uint2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
    }
}

// The same two passes with the coefficients formed in the kernel from the producers' group statistics (GnSrc, coef_to_lds): a workgroup
// takes `ppw` consecutive pixels of ONE image (grid (chunks, N)).  OUT16 = 0: dense fp32 (k_gn_apply), 1: 16-bit (k_gn_apply_h16).
template <int OUT16>
__global__ __launch_bounds__(256) void k_gn_apply_gs(const float *__restrict__ x, long pitch, int HW, int C, const GnSrc gn, int N, int act,
                                                     void *__restrict__ yv, int f16, int ppw) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float *sA = sh, *sB = sh + C;
    const int n = blockIdx.y, tid = threadIdx.x;
    const int cq = C >> 2;
    const int p0 = blockIdx.x * ppw, p1 = min(HW, p0 + ppw);
    const int nq = (p1 - p0) * cq;
    const float *xb = x + ((long)n * HW + p0) * pitch;
    // four quads per thread in flight; the first four are requested before the coefficients are formed (their round trips overlap)
    f32x4 v[4];
    auto fetch = [&](int i0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            if (i < nq) { const int px = i / cq; v[k] = *reinterpret_cast<const f32x4 *>(xb + (long)px * pitch + (i - px * cq) * 4); }
        }
    };
    fetch(tid);
    coef_to_lds(nullptr, nullptr, gn, N, n, sA, sB, sh + 2 * C, tid, 256);
    for (int i0 = tid; i0 < nq; i0 += 1024) {
        if (i0 != tid) fetch(i0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            if (i >= nq) break;
            const int px = i / cq, c = (i - px * cq) * 4;
            const long pix = (long)n * HW + p0 + px;
            const f32x4 a = *reinterpret_cast<const f32x4 *>(sA + c);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(sB + c);
            f32x4 o = v[k] * a + b;
            if (act) { o[0] = silu_f(o[0]); o[1] = silu_f(o[1]); o[2] = silu_f(o[2]); o[3] = silu_f(o[3]); }
            if (OUT16 == 0) {
                *reinterpret_cast<f32x4 *>(static_cast<float *>(yv) + pix * C + c) = o;
            } else if (f16 == 2) {
                store_h2_planes(static_cast<unsigned short *>(yv), pix * C + c, (long)N * HW * C, o);
            } else {
                unsigned short h[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (f16) {
                        const _Float16 hh = (_Float16)o[e];
                        h[e] = __builtin_bit_cast(unsigned short, hh);
                    } else {
                        const unsigned u = __float_as_uint(o[e]);
                        h[e] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
                    }
                }
                *reinterpret_cast<uint2 *>(static_cast<unsigned short *>(yv) + pix * C + c) = uint2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
            }
        }
    }
}
// pixels per workgroup of that pass: about one round of workgroups (8 per compute unit) over the launch, at least four quads per thread
static inline int gn_gs_ppw(int HW, int N, int C) { return std::max(std::max(1, 1024 / (C / 4)), (int)(((long)HW * N + 2047) / 2048)); }

// split-K epilogue: sum the slabs in a fixed order (deterministic), then bias / residual / second output
__global__ void k_splitk_finish(const ConvK p, int splits) {
    const long total = p.M * p.Cout;
    const int hw = p.Hout * p.Wout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / p.Cout;
        const int n = (int)(i - m * p.Cout);
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += p.partial[(long)z * total + i];
        v += p.bias ? p.bias[n] : 0.f;
        if (p.res) v += p.res[m * p.res_pitch + n];
        if (p.out_nchw) {
            const long img = m / hw, rem = m - img * hw;
            p.out[(img * p.Cout + n) * hw + rem] = v;
        } else {
            p.out[m * p.out_pitch + n] = v;
        }
        if (p.out2) p.out2[m * p.out2_pitch + n] = v + p.res2[m * p.res2_pitch + n];
    }
}

// The same, organised for the GroupNorm statistics of the result: a workgroup owns one slot of 32 consecutive pixels x 64 output
// channels (grid (M/32, Cout/64)); its 4 waves take 8 pixels each, a lane one channel; the four partial (sum, sumsq) pairs of a
// channel meet in LDS in a fixed order.  Same arithmetic and order of the slab sum as k_splitk_finish.
__global__ __launch_bounds__(256) void k_splitk_finish_st(const ConvK p, int splits) {
    __shared__ float red[2][4][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + lane;
    const long total = p.M * p.Cout;
    float s1 = 0.f, q1 = 0.f, s2 = 0.f, q2 = 0.f;
    if (n < p.Cout) {
        const float bs = p.bias ? p.bias[n] : 0.f;
        float v[8], r1[8], r2[8];
        const long mb = (long)blockIdx.x * 32 + wave * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        for (int z = 0; z < splits; ++z)
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += p.partial[(long)z * total + (mb + k) * p.Cout + n];
#pragma unroll
        for (int k = 0; k < 8; ++k) { r1[k] = p.res ? p.res[(mb + k) * p.res_pitch + n] : 0.f; r2[k] = p.out2 ? p.res2[(mb + k) * p.res2_pitch + n] : 0.f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] += bs;
            if (p.res) v[k] += r1[k];
            p.out[(mb + k) * p.out_pitch + n] = v[k];
            s1 += v[k]; q1 += v[k] * v[k];
            if (p.out2) {
                const float w = v[k] + r2[k];
                p.out2[(mb + k) * p.out2_pitch + n] = w;
                s2 += w; q2 += w * w;
            }
        }
    }
    red[0][wave][0][lane] = s1; red[0][wave][1][lane] = q1;
    red[1][wave][0][lane] = s2; red[1][wave][1][lane] = q2;
    __syncthreads();
    if (wave < 2) {   // (wave-uniform: wave 0 finishes out's statistics, wave 1 out2's; its 64 lanes = 64 consecutive channels)
        float *st = wave == 0 ? p.st1 : p.st2;
        if (st) {
            const float s = ((red[wave][0][0][lane] + red[wave][1][0][lane]) + red[wave][2][0][lane]) + red[wave][3][0][lane];
            const float q = ((red[wave][0][1][lane] + red[wave][1][1][lane]) + red[wave][2][1][lane]) + red[wave][3][1][lane];
            stat_add_run(st, p.N, ((long)blockIdx.x * 32) / ((long)p.Hout * p.Wout), ((wave == 0 ? p.st1_c0 : p.st2_c0) + n) / (wave == 0 ? p.st1_cg : p.st2_cg),
                         n < p.Cout, (long)p.Hout * p.Wout, s, q);
        }
    }
}

// tf (backward-data of the training path): the logical weight is W'[o][c][ky][kx] = W[c][o][ks-1-ky][ks-1-kx] of a source laid out
// (Cin, Cout, ks, ks) - the flipped, channel-transposed kernel - read in place, no flipped copy is ever materialised
__global__ void k_pack_conv(const float *__restrict__ w, int Cout, int Cin, int Cin_pad, int ks, int rows, float *__restrict__ dst, int tf) {
    const int taps = ks * ks;
    const long Ktot = (long)Cin_pad * taps;
    const long n = (long)rows * Ktot;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i / Ktot);
        const long k = i - (long)o * Ktot;
        const int c16 = (int)(k & 15);
        const long t = k >> 4;
        int cc, tap;
        kt_decode((int)t, Cin_pad >> 4, taps, cc, tap);         // the order k_conv walks K
        const int cin = cc * 16 + c16;
        float v = 0.f;
        if (o < Cout && cin < Cin) v = tf ? w[((long)cin * Cout + o) * taps + (taps - 1 - tap)] : w[((long)o * Cin + cin) * taps + tap];
        dst[i] = v;
    }
}

// weights -> three truncated-bf16 planes (w = hi + mid + lo exactly), laid out as k_conv_bf3 stages them
__global__ void k_pack_conv_bf3(const float *__restrict__ w, int Cout, int Cin, int Cin_pad, int ks, int rows,
                                unsigned short *__restrict__ dst, int tf) {
    const int taps = ks * ks;
    const long Ktot = (long)Cin_pad * taps;
    const long n = (long)rows * Ktot;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i / Ktot);
        const long k = i - (long)o * Ktot;
        const int c16 = (int)(k & 15);
        const long t = k >> 4;
        int cc, tap;
        kt_decode((int)t, Cin_pad >> 4, taps, cc, tap);
        const int cin = cc * 16 + c16;
        float v = 0.f;
        if (o < Cout && cin < Cin) v = tf ? w[((long)cin * Cout + o) * taps + (taps - 1 - tap)] : w[((long)o * Cin + cin) * taps + tap];
        const unsigned uh = __float_as_uint(v) & 0xffff0000u;
        const float r1 = v - __uint_as_float(uh);
        const unsigned um = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(um);
        const unsigned ul = __float_as_uint(r2);
        unsigned short *q = dst + ((long)o * (Ktot >> 4) + t) * 48 + (c16 >> 3) * 8 + (c16 & 7);
        q[0] = (unsigned short)(uh >> 16);
        q[16] = (unsigned short)(um >> 16);
        q[32] = (unsigned short)(ul >> 16);
    }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics
// ---------------------------------------------------------------------------------------------
// grid (nchunks, N); blockDim = (C/4) * k threads; thread owns one float4 channel column.
__global__ void k_gn_partial(const float *__restrict__ x, long pitch, int HW, int C, int nchunks, float *__restrict__ partial,
                             const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ emb,
                             long emb_pitch, float *__restrict__ cA, float *__restrict__ cB, float *__restrict__ gstat, float eps) {
    extern __shared__ float sh[];  // [k][C] sums then [k][C] sumsq
    const int cq = C >> 2;
    const int k = blockDim.x / cq;
    const int c4 = threadIdx.x % cq, prow = threadIdx.x / cq;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int per = (HW + nchunks - 1) / nchunks;
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, ss = f32x4{0.f, 0.f, 0.f, 0.f};
    if (prow < k) {
        const float *base = x + (long)n * HW * pitch + c4 * 4;
        int pp = p0 + prow;
        for (; pp + 3 * k < p1; pp += 4 * k) {   // four independent loads in flight
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(base + (long)pp * pitch);
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(base + (long)(pp + k) * pitch);
            const f32x4 v2 = *reinterpret_cast<const f32x4 *>(base + (long)(pp + 2 * k) * pitch);
            const f32x4 v3 = *reinterpret_cast<const f32x4 *>(base + (long)(pp + 3 * k) * pitch);
            s += v0; ss += v0 * v0;
            s += v1; ss += v1 * v1;
            s += v2; ss += v2 * v2;
            s += v3; ss += v3 * v3;
        }
        for (; pp < p1; pp += k) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (long)pp * pitch);
            s += v;
            ss += v * v;
        }
        float *d = sh + (long)prow * C + c4 * 4;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
        float *d2 = sh + (long)(k + prow) * C + c4 * 4;
        d2[0] = ss[0]; d2[1] = ss[1]; d2[2] = ss[2]; d2[3] = ss[3];
    }
    __syncthreads();
    const int cg = C / 32;
    __shared__ float gs[64];
    if (threadIdx.x < 64) {
        const int g = threadIdx.x & 31, which = threadIdx.x >> 5;
        float t = 0.f;
        for (int r = 0; r < k; ++r)
            for (int c = 0; c < cg; ++c) t += sh[(long)(which * k + r) * C + g * cg + c];
        partial[(((long)n * nchunks + chunk) * 32 + g) * 2 + which] = t;
        gs[which * 32 + g] = t;
    }
    if (nchunks != 1 || cA == nullptr) return;
    // one workgroup saw the whole image: finish the affine here (saves the k_gn_coef launch)
    __syncthreads();
    const double cnt = (double)HW * cg;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cg;
        const double mean = (double)gs[g] / cnt;
        double var = (double)gs[32 + g] / cnt - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        float a = rstd * gamma[c];
        float b = beta[c] - (float)mean * a;
        if (emb) {
            const float sc = 1.f + emb[(long)n * emb_pitch + c];
            const float sf = emb[(long)n * emb_pitch + C + c];
            a = a * sc;
            b = b * sc + sf;
        }
        cA[(long)n * C + c] = a;
        cB[(long)n * C + c] = b;
        if (gstat && c % cg == 0) { gstat[((long)n * 32 + c / cg) * 2] = (float)mean; gstat[((long)n * 32 + c / cg) * 2 + 1] = rstd; }
    }
}

// Small tensors (HW*C <= 512K): one 256-thread workgroup per (group, image) reads the group's HW x C/32 slab and
// writes the group's affine directly - 32*N workgroups in flight instead of N, no second launch.
template <int V>
__global__ __launch_bounds__(256) void k_gn_small(const float *__restrict__ x, long pitch, int HW, int C,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                  const float *__restrict__ emb, long emb_pitch, float *__restrict__ cA,
                                                  float *__restrict__ cB, float *__restrict__ gstat, float eps) {
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cg = C / 32, per = cg / V;
    const float *base = x + (long)n * HW * pitch + g * cg;
    float s = 0.f, ss = 0.f;
    const int items = HW * per;
    for (int i = tid; i < items; i += 256) {
        const int pix = i / per, sub = i - pix * per;
        const float *q = base + (long)pix * pitch + sub * V;
        if (V == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(q);
            s += (v[0] + v[1]) + (v[2] + v[3]);
            ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        } else if (V == 2) {
            const float2 v = *reinterpret_cast<const float2 *>(q);
            s += v.x + v.y;
            ss += v.x * v.x + v.y * v.y;
        } else {
            const float v = *q;
            s += v;
            ss += v * v;
        }
    }
    double ds = (double)s, dss = (double)ss;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        ds += __shfl_xor(ds, d);
        dss += __shfl_xor(dss, d);
    }
    __shared__ double red[8];
    if ((tid & 63) == 0) { red[(tid >> 6) * 2] = ds; red[(tid >> 6) * 2 + 1] = dss; }
    __syncthreads();
    ds = (red[0] + red[2]) + (red[4] + red[6]);
    dss = (red[1] + red[3]) + (red[5] + red[7]);
    const double cnt = (double)HW * cg;
    const double mean = ds / cnt;
    double var = dss / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int j = tid; j < cg; j += 256) {
        const int c = g * cg + j;
        float a = rstd * gamma[c];
        float b = beta[c] - (float)mean * a;
        if (emb) {
            const float sc = 1.f + emb[(long)n * emb_pitch + c];
            const float sf = emb[(long)n * emb_pitch + C + c];
            a = a * sc;
            b = b * sc + sf;
        }
        cA[(long)n * C + c] = a;
        cB[(long)n * C + c] = b;
        if (gstat && c % cg == 0) { gstat[((long)n * 32 + c / cg) * 2] = (float)mean; gstat[((long)n * 32 + c / cg) * 2 + 1] = rstd; }
    }
}

// grid (32 groups, N), one wave each: lanes sum the chunk partials (double), then lanes < C/32 write
// A = rstd*gamma [*(1+scale)], B = (beta - mean*rstd*gamma) [*(1+scale) + shift] for the group's channels
__global__ __launch_bounds__(64) void k_gn_coef(const float *__restrict__ partial, int nchunks, int HW, int C,
                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                const float *__restrict__ emb, long emb_pitch, float *__restrict__ cA,
                                                float *__restrict__ cB, float *__restrict__ gstat, float eps) {
    const int g = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
    const int cg = C / 32;
    double s = 0.0, ss = 0.0;
    for (int k = lane; k < nchunks; k += 64) {
        const float *pp = partial + (((long)n * nchunks + k) * 32 + g) * 2;
        s += (double)pp[0];
        ss += (double)pp[1];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s += __shfl_xor(s, d);
        ss += __shfl_xor(ss, d);
    }
    const double cnt = (double)HW * cg;
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int j = lane; j < cg; j += 64) {
        const int c = g * cg + j;
        float a = rstd * gamma[c];
        float b = beta[c] - (float)mean * a;
        if (emb) {
            const float sc = 1.f + emb[(long)n * emb_pitch + c];
            const float sh = emb[(long)n * emb_pitch + C + c];
            a = a * sc;
            b = b * sc + sh;
        }
        cA[(long)n * C + c] = a;
        cB[(long)n * C + c] = b;
        if (gstat && c % cg == 0) { gstat[((long)n * 32 + c / cg) * 2] = (float)mean; gstat[((long)n * 32 + c / cg) * 2 + 1] = rstd; }
    }
}

// GroupNorm affine (as arrays) from the fixed-point group totals the producing kernels left (ConvK::st1 / st2, hl_stats.h) - the tensor
// itself is not read again.  grid (32 groups, N), one wave; same arithmetic as coef_to_lds.
__global__ __launch_bounds__(64) void k_gn_coef_tot(const float *__restrict__ gt, int HW, int C, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, const float *__restrict__ emb, long emb_pitch,
                                                    float *__restrict__ cA, float *__restrict__ cB, float eps) {
    const int g = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
    const int cg = C / 32, c_lo = g * cg;
    float mean, rstd;
    group_mean_rstd(gt, gridDim.y, n, g, HW, cg, eps, mean, rstd);
    for (int j = lane; j < cg; j += 64) {
        const int c = c_lo + j;
        float a = rstd * gamma[c];
        float b = beta[c] - mean * a;
        if (emb) {
            const float sc = 1.f + emb[(long)n * emb_pitch + c];
            const float sf = emb[(long)n * emb_pitch + C + c];
            a = a * sc;
            b = b * sc + sf;
        }
        cA[(long)n * C + c] = a;
        cB[(long)n * C + c] = b;
    }
}

// Totals of a tensor nobody left totals for (the single-convolution entry points): sum and sum of squares of image blockIdx.y's values, every
// workgroup adding its slice to "group" blockIdx.x & 31 - the fp16x2 kernels only need the image's sum x^2 (act_scale_totals), any grouping serves.
__global__ __launch_bounds__(256) void k_tensor_totals(const float *__restrict__ x, long pitch, long HW, int C, float *__restrict__ st) {
    __shared__ float rs[256], rq[256];
    const int n = blockIdx.y, N = gridDim.y;
    const long per = (HW + gridDim.x - 1) / gridDim.x, p0 = (long)blockIdx.x * per, p1 = p0 + per < HW ? p0 + per : HW;
    float s = 0.f, q = 0.f;
    const int c4 = C / 4;
    for (long i = (p0 * c4) + threadIdx.x; i < p1 * c4; i += 256) {
        const long pix = i / c4;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + ((long)n * HW + pix) * pitch + (i - pix * c4) * 4);
        s += (v[0] + v[1]) + (v[2] + v[3]);
        q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    rs[threadIdx.x] = s; rq[threadIdx.x] = q;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) { rs[threadIdx.x] += rs[threadIdx.x + d]; rq[threadIdx.x] += rq[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0 && p1 > p0) stat_add(st, N, n, blockIdx.x & 31, HW, rs[0], rq[0]);
}

// the largest |x| of image blockIdx.y (tensor_absmax): what bounds the power-of-two scale of a raw input of the fp16x2 kernels exactly, whatever the magnitude
__global__ __launch_bounds__(256) void k_tensor_absmax(const float *__restrict__ x, long pitch, long HW, int C, float *__restrict__ amax) {
    __shared__ float rm[256];
    const int n = blockIdx.y;
    const long per = (HW + gridDim.x - 1) / gridDim.x, p0 = (long)blockIdx.x * per, p1 = p0 + per < HW ? p0 + per : HW;
    float m = 0.f;
    const int c4 = C / 4;
    for (long i = (p0 * c4) + threadIdx.x; i < p1 * c4; i += 256) {
        const long pix = i / c4;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + ((long)n * HW + pix) * pitch + (i - pix * c4) * 4);
        // (integer max of the magnitudes' bit patterns: a NaN stays the largest and reaches the consumer, whose pow2_scale_for_bound then keeps scale 1)
        const unsigned a0 = __float_as_uint(v[0]) & 0x7fffffffu, a1 = __float_as_uint(v[1]) & 0x7fffffffu, a2 = __float_as_uint(v[2]) & 0x7fffffffu, a3 = __float_as_uint(v[3]) & 0x7fffffffu;
        m = __uint_as_float(max(max(__float_as_uint(m), a0), max(max(a1, a2), a3)));
    }
    rm[threadIdx.x] = m;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) rm[threadIdx.x] = __uint_as_float(max(__float_as_uint(rm[threadIdx.x]), __float_as_uint(rm[threadIdx.x + d])));
        __syncthreads();
    }
    if (threadIdx.x == 0 && p1 > p0) atomicMax(reinterpret_cast<unsigned *>(amax) + n, __float_as_uint(rm[0]));
}

// ---------------------------------------------------------------------------------------------
// small-batch linear: one wave per output row
// ---------------------------------------------------------------------------------------------
template <int MAXB>
__global__ __launch_bounds__(256) void k_linear_small(const float *__restrict__ in, long in_pitch, int B, int K,
                                                      const float *__restrict__ W, const float *__restrict__ bias, int O,
                                                      int silu_in, const float *__restrict__ addrow,
                                                      const int64_t *__restrict__ idx, float *__restrict__ out, long out_pitch) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= O) return;
    float acc[MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) acc[b] = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(long)o * K + k];
#pragma unroll
        for (int b = 0; b < MAXB; ++b)
            if (b < B) {
                float v = in[(long)b * in_pitch + k];
                if (silu_in) v = silu_f(v);
                acc[b] = fmaf(v, w, acc[b]);
            }
    }
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        float v = acc[b];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
        if (lane == 0 && b < B) {
            v += bias ? bias[o] : 0.f;
            if (addrow) v += addrow[(long)idx[b] * O + o];
            out[(long)b * out_pitch + o] = v;
        }
    }
}

__global__ void k_timestep_embedding(const int64_t *__restrict__ t, const float *__restrict__ tf, int B, int dim, float *__restrict__ out) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, j = i - b * dim;
    float v = 0.f;
    if (j < 2 * half) {
        const int k = j < half ? j : j - half;
        // freqs = exp(-ln(10000) * k / half) in fp32 (nn.py:113-116)
        const float f = expf(-9.210340371976184f * (float)k / (float)half);
        const float a = (tf ? tf[b] : (float)t[b]) * f;
        v = j < half ? cosf(a) : sinf(a);
    }
    out[i] = v;
}

// ---------------------------------------------------------------------------------------------
// attention: one wave = 32 queries of one (n, head); 4 waves per block share K/V tiles in LDS.
// S^T[key][q] = K[key][:] . Q^T[:, q]  (A = K tile from LDS, B = Q^T from registers)
// O^T[c][q]  += V^T[c][key] . P^T[key][q]  (A = V tile from LDS read "down the keys", B = P^T = the
// S^T accumulator itself after softmax - same register-resident trick as the render MLP)
// ---------------------------------------------------------------------------------------------
template <int CH, int WPB>
__global__ __launch_bounds__(WPB * 64, 1) void k_attention(const float *__restrict__ qkv, int T, int C, int heads, float *__restrict__ out) {
    constexpr int CT = CH / 32;   // channel tiles of the output
    constexpr int KS = CH / 2;    // k-steps of the QK^T product
    constexpr int LDK = CH + 1;   // odd row stride: rows differ per lane in the A reads
    __shared__ float sK[32 * LDK];
    __shared__ __attribute__((aligned(16))) float sV[32 * CH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int nh = blockIdx.y;  // n*heads + head
    const int n = nh / heads, head = nh % heads;
    const int q0 = (blockIdx.x * WPB + wave) * 32;
    const float scale = 1.f / sqrtf(sqrtf((float)CH));
    const long pitch = 3L * C;
    const float *base = qkv + (long)n * T * pitch + (long)head * 3 * CH;

    // Q^T operand: lane (query j, half) holds q[c = 2s + half] * scale
    const int qj = min(q0 + (lane & 31), T - 1);
    float qreg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qreg[s] = base[(long)qj * pitch + 2 * s + half] * scale;

    f32x16 o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    float mrun = -3.0e38f, lrun = 0.f;

    for (int k0 = 0; k0 < T; k0 += 32) {
        __syncthreads();
        for (int e = tid; e < 32 * (CH / 4); e += WPB * 64) {   // 16-byte global loads; K rows keep an odd LDS stride
            const int key = e / (CH / 4), c = (e - key * (CH / 4)) * 4;
            const int kk = min(k0 + key, T - 1);
            const f32x4 kv = *reinterpret_cast<const f32x4 *>(base + (long)kk * pitch + CH + c);
            const f32x4 vv = *reinterpret_cast<const f32x4 *>(base + (long)kk * pitch + 2 * CH + c);
            float *dk = sK + key * LDK + c;
            dk[0] = kv[0] * scale; dk[1] = kv[1] * scale; dk[2] = kv[2] * scale; dk[3] = kv[3] * scale;
            *reinterpret_cast<f32x4 *>(sV + key * CH + c) = vv;
        }
        __syncthreads();
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[(lane & 31) * LDK + 2 * s + half], qreg[s], st, 0, 0, 0);
        // st[r] = score(key = (r&3)+8*(r>>2)+4*half, query = lane&31); mask keys beyond T
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key >= T) st[r] = -3.0e38f;
            mx = fmaxf(mx, st[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = __expf(st[r] - mnew);
            psum += st[r];
        }
        psum += __shfl_xor(psum, 32);
        lrun = lrun * alpha + psum;
        mrun = mnew;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[c][r] *= alpha;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int key = (s & 3) + 8 * (s >> 2) + 4 * half;  // the key this lane's st[s] belongs to
                o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[key * CH + c * 32 + (lane & 31)], st[s], o[c], 0, 0, 0);
            }
        }
    }
    // o[c][r] = O^T[channel = c*32 + (r&3)+8*(r>>2)+4*half][query = lane&31]
    const int qi = q0 + (lane & 31);
    if (qi < T) {
        const float inv = 1.f / lrun;
        float *dst = out + ((long)n * T + qi) * C + head * CH;
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[c * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = o[c][r] * inv;
    }
}

// Key-split variant for short sequences (the UNet's 32x32 / 16x16 / 8x8 attention levels), where one wave per
// 32-query tile leaves most of the 1024 SIMDs idle: the KW waves of a workgroup share ONE query tile and take every
// KW-th key tile each, then merge their (max, sum, O) partials through LDS.  No workgroup barrier inside the key
// loop: K rows go straight from global memory into the MFMA A operand (lane = key, the two halves of the wave take
// alternate groups of 4 channels; Q is loaded with the same permutation), V tiles are staged in a wave-private LDS
// region, and with PF the next tile's K and V are in flight (registers) while the current one is multiplied.
// eight values (two f32x4, scaled by `sc`) -> the two fp16 planes of one 32x32x16 operand (round 6: h0 = the nearest fp16, h1 = the nearest fp16 of the residual - split_h2 of
// hl_conv_h16.hip: 2^-24 while both planes are normal, and a value beyond fp16's range becomes inf / NaN instead of saturating)
// (round 6, measured: moving K's `scale` factor to Q - so that K goes into the split as loaded, sc = 1.f - produces GARBAGE scores in k_attention_ks<96, 4, true, true>
//  with this compiler, while sc = 1.00001f is exact to 1e-5 as it should be: the kernel keeps K's tile in accumulator registers between iterations, and the asm statement
//  fed straight from those copies is miscompiled.  Every use below has a vector instruction between the load / copy and the asm, and the tests pin the results bit for bit.)
__device__ __forceinline__ void att_split8(const f32x4 a, const f32x4 b, float sc, u32x4 &p0, u32x4 &p1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x = (q < 2 ? a[2 * q] : b[2 * q - 4]) * sc, y = (q < 2 ? a[2 * q + 1] : b[2 * q - 3]) * sc;
        unsigned w0, w1;
        hl_split2_rne(x, y, w0, w1);
        p0[q] = w0; p1[q] = w1;
    }
}
typedef _Float16 att_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 att_mma(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(att_f16x8, a), __builtin_bit_cast(att_f16x8, b), c, 0, 0, 0);
}

// H2 (round 5, the default mode): both products - scores K Q^T and O^T += V^T P - from fp16x2 operands on v_mfma_f32_32x32x16_f16 (two fp16 planes per operand, three partial
// products, fp32 accumulation; the convolutions' scheme): a k-step of the scores = two of the 8-channel groups (the lane half's 2 x 4 channels of K against the same channels
// of Q, split once per query tile); a k-step of O^T = 8 of the lane half's 16 keys - the probabilities are the score accumulators split in place, V comes from the wave's LDS
// tile as before (one dword per key and lane).  36 (CH = 96) / 72 (CH = 192) MFMAs of 32 cycles per key tile instead of 96 / 192 of 64.
template <int CH, int KW, bool PF, bool H2>
__global__ __launch_bounds__(KW * 64, 1) void k_attention_ks(const float *__restrict__ qkv, int T, int C, int heads,
                                                             float *__restrict__ out, float *__restrict__ out_tot, int N) {
    constexpr int CT = CH / 32, NG = CH / 8, WLDS = CH * 33 + 64;
    extern __shared__ __attribute__((aligned(16))) float sh[];   // [KW][WLDS]: V tile (32 x CH), later O^T (CH x 33) + m,l
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *sV = sh + wave * WLDS;
    const int nh = blockIdx.y, n = nh / heads, head = nh % heads;
    const int q0 = blockIdx.x * 32;
    const float scale = 1.f / sqrtf(sqrtf((float)CH));
    const long pitch = 3L * C;
    const float *base = qkv + (long)n * T * pitch + (long)head * 3 * CH;

    // contraction order: MFMA (m, i) multiplies channel 8m + 4*half + i
    const int qj = min(q0 + (lane & 31), T - 1);
    f32x4 qreg[NG];
#pragma unroll
    for (int m = 0; m < NG; ++m) {
        qreg[m] = *reinterpret_cast<const f32x4 *>(base + (long)qj * pitch + 8 * m + 4 * half);
#pragma unroll
        for (int i = 0; i < 4; ++i) qreg[m][i] *= scale;
    }
    u32x4 qp[H2 ? NG / 2 : 1][2];                        // (H2) Q as two fp16 planes per k-step: channels of groups 2j, 2j + 1
    if constexpr (H2) {
#pragma unroll
        for (int j = 0; j < NG / 2; ++j) att_split8(qreg[2 * j], qreg[2 * j + 1], 1.f, qp[j][0], qp[j][1]);
    }
    auto loadK = [&](int k0, f32x4(&kr)[NG]) {
        const int kk = min(k0 + (lane & 31), T - 1);
#pragma unroll
        for (int m = 0; m < NG; ++m) kr[m] = *reinterpret_cast<const f32x4 *>(base + (long)kk * pitch + CH + 8 * m + 4 * half);
    };
    auto loadV = [&](int k0, f32x4(&vr)[NG]) {   // 32 keys x CH floats = NG float4 per lane, consecutive lanes along a row
#pragma unroll
        for (int e = 0; e < NG; ++e) {
            const int idx = e * 64 + lane, key = idx / (CH / 4), c = (idx - key * (CH / 4)) * 4;
            vr[e] = *reinterpret_cast<const f32x4 *>(base + (long)min(k0 + key, T - 1) * pitch + 2 * CH + c);
        }
    };
    // H2 (round 6): the V tile is stored multiplied by a power of two `vs` that puts the largest |v| the wave has seen so far just below 2^15 - the two fp16 planes of a value
    // are both normal only above 2^-3, and V is whatever the qkv convolution produced (V x 2^-10: rel-L2 1.8e-5 against float64 without this, 3.6e-7 with).  The scale only
    // ever shrinks; when it does, the accumulated O is multiplied by the ratio together with the softmax's own rescaling factor (a power of two: exact), and O is divided
    // by it once before the waves' partials are merged.  A tile of zeros, or one with a value that is not finite, leaves the scale alone.
    float vs = 3.0e38f, vs_ratio = 1.f;     // (3.0e38: no tile seen yet)
    auto storeV = [&](f32x4(&vr)[NG]) {      // (H2: scales vr in place - no second copy of the tile in registers)
        if constexpr (H2) {
            float tm = 0.f;
#pragma unroll
            for (int e = 0; e < NG; ++e) {      // (two v_max3_f32 with |.| operands per four values)
                tm = fmaxf(fmaxf(fabsf(vr[e][0]), fabsf(vr[e][1])), tm);
                tm = fmaxf(fmaxf(fabsf(vr[e][2]), fabsf(vr[e][3])), tm);
            }
            const float wm = __uint_as_float(wave_max_u32(__float_as_uint(tm)));      // (non-negative floats order like their bit patterns; a NaN is the largest)
            float vt = pow2_scale_for_bound(wm);                                     // 1 for 0 / inf / NaN
            if (!(wm > 0.f) || !(wm < 3.0e38f)) vt = vs < 3.0e38f ? vs : 1.f;
            vs_ratio = 1.f;
            if (vt < vs) { vs_ratio = vs < 3.0e38f ? vt / vs : 1.f; vs = vt; }
        }
#pragma unroll
        for (int e = 0; e < NG; ++e) {
            const int idx = e * 64 + lane, key = idx / (CH / 4), c = (idx - key * (CH / 4)) * 4;
            if constexpr (H2) vr[e] *= vs;
            *reinterpret_cast<f32x4 *>(sV + key * CH + c) = vr[e];
        }
    };

    f32x16 o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    float mrun = -3.0e38f, lrun = 0.f;

    f32x4 kc[NG], vn[NG], kn[NG];   // kn unused (and eliminated) without PF
    int k0 = wave * 32;
    if (k0 < T) { loadK(k0, kc); loadV(k0, vn); }
    for (; k0 < T; k0 += KW * 32) {
        storeV(vn);
        const int k1 = k0 + KW * 32;
        if constexpr (PF) { if (k1 < T) { loadK(k1, kn); loadV(k1, vn); } }
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        if constexpr (H2) {
#pragma unroll
            for (int j = 0; j < NG / 2; ++j) {
                u32x4 k0p, k1p;
                att_split8(kc[2 * j], kc[2 * j + 1], scale, k0p, k1p);
                st = att_mma(k1p, qp[j][0], st);           // smallest partial product first
                st = att_mma(k0p, qp[j][1], st);
                st = att_mma(k0p, qp[j][0], st);
            }
        } else {
#pragma unroll
            for (int m = 0; m < NG; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[m][i] * scale, qreg[m][i], st, 0, 0, 0);
        }
        // st[r] = score(key = (r&3)+8*(r>>2)+4*half, query = lane&31); mask keys beyond T
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key >= T) st[r] = -3.0e38f;
            mx = fmaxf(mx, st[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = __expf(st[r] - mnew);
            psum += st[r];
        }
        psum += __shfl_xor(psum, 32);
        lrun = lrun * alpha + psum;
        mrun = mnew;
        if constexpr (H2) {
            u32x4 pp[2][2];                                  // the probabilities of this lane half's keys st[8 jj .. 8 jj + 7] as two planes
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                att_split8(f32x4{st[8 * jj], st[8 * jj + 1], st[8 * jj + 2], st[8 * jj + 3]}, f32x4{st[8 * jj + 4], st[8 * jj + 5], st[8 * jj + 6], st[8 * jj + 7]}, 1.f, pp[jj][0], pp[jj][1]);
            const float alpha_o = alpha * vs_ratio;         // (O carries the V scale: both factors at once)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[c][r] *= alpha_o;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    f32x4 va, vb;                            // V of this lane's channel at the same eight keys, in the same order
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        va[i] = sV[(i + 16 * jj + 4 * half) * CH + c * 32 + (lane & 31)];
                        vb[i] = sV[(i + 8 + 16 * jj + 4 * half) * CH + c * 32 + (lane & 31)];
                    }
                    u32x4 v0p, v1p;
                    att_split8(va, vb, 1.f, v0p, v1p);
                    o[c] = att_mma(v1p, pp[jj][0], o[c]);
                    o[c] = att_mma(v0p, pp[jj][1], o[c]);
                    o[c] = att_mma(v0p, pp[jj][0], o[c]);
                }
            }
        } else {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[c][r] *= alpha;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int key = (s & 3) + 8 * (s >> 2) + 4 * half;  // the key this lane's st[s] belongs to
                o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[key * CH + c * 32 + (lane & 31)], st[s], o[c], 0, 0, 0);
            }
        }
        }
        if constexpr (PF) {
#pragma unroll
            for (int m = 0; m < NG; ++m) kc[m] = kn[m];
        } else if (k1 < T) {
            loadK(k1, kc);
            loadV(k1, vn);
        }
    }

    // merge the KW partials: O^T of every wave to its LDS region as [channel][33] (+ running max / sum per query)
    __syncthreads();
    const float vinv = (H2 && vs < 3.0e38f) ? 1.f / vs : 1.f;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) sV[(c * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 33 + (lane & 31)] = o[c][r] * vinv;
    if (half == 0) {
        sV[CH * 33 + lane] = mrun;
        sV[CH * 33 + 32 + lane] = lrun;
    }
    __syncthreads();
    float osq = 0.f;
    for (int e = tid; e < 32 * CH; e += KW * 64) {
        const int q = e / CH, c = e - q * CH;
        float ms = -3.0e38f;
#pragma unroll
        for (int w = 0; w < KW; ++w) ms = fmaxf(ms, sh[w * WLDS + CH * 33 + q]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < KW; ++w) {
            const float f = __expf(sh[w * WLDS + CH * 33 + q] - ms);
            num += sh[w * WLDS + c * 33 + q] * f;
            den += sh[w * WLDS + CH * 33 + 32 + q] * f;
        }
        if (q0 + q < T) {
            const float val = num / den;
            out[((long)n * T + q0 + q) * C + head * CH + c] = val;
            osq += val * val;
        }
    }
    // sum x^2 of what this workgroup stored, into the fixed-point totals of the output (any "group": act_scale_totals adds all 32) - the projection convolution behind
    // the attention reads its raw input's power-of-two scale from them (hl_stats.h), as every other raw-input convolution does from its producers' totals
    if (out_tot) {
        osq = wave_sum_f32(osq);
        __syncthreads();
        if (lane == 0) sh[wave] = osq;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < KW; ++w) t += sh[w];
            stat_add(out_tot, N, n, (int)(blockIdx.x & 31), (long)T, 0.f, t);
        }
    }
}

// generic (slow) attention for head widths that are not a multiple of 32: one thread per (query, channel-chunk)
__global__ void k_attention_generic(const float *__restrict__ qkv, int T, int C, int heads, int ch, float *__restrict__ out) {
    extern __shared__ float sh[];  // scores of one query row per wave: [waves][T]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nh = blockIdx.y, n = nh / heads, head = nh % heads;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= T) return;
    const float scale = 1.f / sqrtf(sqrtf((float)ch));
    const long pitch = 3L * C;
    const float *base = qkv + (long)n * T * pitch + (long)head * 3 * ch;
    float *sc = sh + (long)wave * T;
    float mx = -3.0e38f;
    for (int k = lane; k < T; k += 64) {
        float s = 0.f;
        for (int c = 0; c < ch; ++c) s = fmaf(base[(long)qi * pitch + c] * scale, base[(long)k * pitch + ch + c] * scale, s);
        sc[k] = s;
        mx = fmaxf(mx, s);
    }
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    float sum = 0.f;
    for (int k = lane; k < T; k += 64) {
        const float e = __expf(sc[k] - mx);
        sc[k] = e;
        sum += e;
    }
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    __builtin_amdgcn_s_waitcnt(0);
    for (int c = lane; c < ch; c += 64) {
        float a = 0.f;
        for (int k = 0; k < T; ++k) a = fmaf(sc[k], base[(long)k * pitch + 2 * ch + c], a);
        out[((long)n * T + qi) * C + head * ch + c] = a / sum;
    }
}

// grid (workgroups per image, B).  xo_tot / xs_tot (optional, zeroed): the sum x^2 of each image of the two outputs as fixed-point totals - the 27 -> 192 input convolutions
// read their raw input's power-of-two scale from them like every other fp16x2 convolution (hl_stats.h)
__global__ __launch_bounds__(256) void k_prep_inputs(const float *__restrict__ x, const float *__restrict__ xc, int B, int C, int HW, int Cpad,
                                                     float *__restrict__ xo, float *__restrict__ xs, float *__restrict__ xo_tot, float *__restrict__ xs_tot) {
    __shared__ float red[2][4];
    const long b = blockIdx.y, n_img = (long)HW * Cpad;
    float q1 = 0.f, q2 = 0.f;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n_img; j += (long)gridDim.x * blockDim.x) {
        const int c = (int)(j % Cpad);
        const long r = j / Cpad;
        float v = 0.f, s = 0.f;
        if (c < C) {
            v = x[(b * C + c) * HW + r];
            s = xc ? v + xc[(b * C + c) * HW + r] : v;
        }
        xo[b * n_img + j] = v;
        if (xs) xs[b * n_img + j] = s;
        q1 += v * v;
        q2 += s * s;
    }
    if (xo_tot == nullptr) return;
    q1 = wave_sum_f32(q1);
    q2 = wave_sum_f32(q2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = q1; red[1][wave] = q2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        stat_add(xo_tot, B, b, (int)(blockIdx.x & 31), (long)HW, 0.f, ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]);
        if (xs && xs_tot) stat_add(xs_tot, B, b, (int)(blockIdx.x & 31), (long)HW, 0.f, ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]);
    }
}

// k_gn_apply into the channel-blocked layout the F(4x4) convolution reads, y[(c/8)][pixel][c%8]: a workgroup normalises tp pixels x C
// channels with coalesced 16-byte reads, turns them through LDS (row length rs = C + pad, (rs/4) % 16 == 4: the 16-lane groups of
// the read-back hit 16 different slots) and writes tp x 32 contiguous bytes per 8-channel plane
__global__ __launch_bounds__(256) void k_gn_apply_blk(const float *__restrict__ x, long pitch, long pixels_per_img, long npix, int C,
                                                      const float *__restrict__ cA, const float *__restrict__ cB, int act,
                                                      float *__restrict__ y, int tp, int rs, const GnSrc gn, int N, int reps) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    f32x4 *sh4 = reinterpret_cast<f32x4 *>(sh);                      // (indexed in 16-byte units: the compiler then emits ds_*_b128)
    f32x4 *y4 = reinterpret_cast<f32x4 *>(y);
    const int cq = C >> 2, rs4 = rs >> 2, tid = threadIdx.x;
    const long pix0 = (long)blockIdx.x * tp;
    if (cA == nullptr) {   // coefficients from the producers' totals: formed once per workgroup, which then walks `reps` tiles of tp pixels of ONE image
        float *sA = sh + tp * rs, *sB = sA + C;
        const long t0 = (long)blockIdx.x * reps;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f};                  // the thread's first quad of the first tile is requested before the coefficients are formed
        if (tid < tp * cq) v0 = *reinterpret_cast<const f32x4 *>(x + (t0 * tp + tid / cq) * pitch + (tid % cq) * 4);
        coef_to_lds(nullptr, nullptr, gn, N, (int)(t0 * tp / pixels_per_img), sA, sB, sB + C, tid, 256);
        const int per = tp * 2, nkt = C >> 3;
        for (int r = 0; r < reps; ++r) {
            const long pb = (t0 + r) * tp;
            for (int idx = tid; idx < tp * cq; idx += 256) {
                const int px = idx / cq, c4 = idx - px * cq;
                const f32x4 v = (r == 0 && idx == tid) ? v0 : *reinterpret_cast<const f32x4 *>(x + (pb + px) * pitch + c4 * 4);
                f32x4 o = v * *reinterpret_cast<const f32x4 *>(sA + c4 * 4) + *reinterpret_cast<const f32x4 *>(sB + c4 * 4);
                if (act) { o[0] = silu_f(o[0]); o[1] = silu_f(o[1]); o[2] = silu_f(o[2]); o[3] = silu_f(o[3]); }
                sh4[px * rs4 + c4] = o;
            }
            __syncthreads();
            for (int idx = tid; idx < nkt * per; idx += 256) {
                const int kt = idx / per, rem = idx - kt * per, px = rem >> 1, h = rem & 1;
                y4[((long)kt * npix + pb + px) * 2 + h] = sh4[px * rs4 + kt * 2 + h];
            }
            __syncthreads();
        }
        return;
    } else
    for (int idx = tid; idx < tp * cq; idx += 256) {
        const int px = idx / cq, c4 = idx - px * cq;
        const long pix = pix0 + px, n = pix / pixels_per_img;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + pix * pitch + c4 * 4);
        const f32x4 a = *reinterpret_cast<const f32x4 *>(cA + n * C + c4 * 4);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(cB + n * C + c4 * 4);
        f32x4 o = v * a + b;
        if (act) { o[0] = silu_f(o[0]); o[1] = silu_f(o[1]); o[2] = silu_f(o[2]); o[3] = silu_f(o[3]); }
        sh4[px * rs4 + c4] = o;
    }
    __syncthreads();
    const int per = tp * 2, nkt = C >> 3;
    for (int idx = tid; idx < nkt * per; idx += 256) {
        const int kt = idx / per, rem = idx - kt * per, px = rem >> 1, h = rem & 1;
        y4[((long)kt * npix + pix0 + px) * 2 + h] = sh4[px * rs4 + kt * 2 + h];
    }
}

// ---- cond_type='cross_attention' (spatial_transformer.py): LayerNorm over the channels of every token, GEGLU, a per-image row vector ----
// y (npix, C dense) = (x - mean) * rstd * gamma + beta per pixel (nn.LayerNorm(C), eps 1e-5); one wave per pixel
__global__ __launch_bounds__(256) void k_layernorm(const float *__restrict__ x, long pitch, long npix, int C, const float *__restrict__ gamma,
                                                   const float *__restrict__ beta, float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= npix) return;
    const float *xp = x + pix * pitch;
    float sm = 0.f;
    for (int c = lane; c < C; c += 64) sm += xp[c];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sm += __shfl_xor(sm, d);
    const float mean = sm / (float)C;
    float sq = 0.f;
    for (int c = lane; c < C; c += 64) { const float dv = xp[c] - mean; sq = fmaf(dv, dv, sq); }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sq += __shfl_xor(sq, d);
    const float rstd = 1.f / sqrtf(sq / (float)C + 1e-5f);
    for (int c = lane; c < C; c += 64) y[pix * C + c] = fmaf((xp[c] - mean) * rstd, gamma[c], beta[c]);
}
// GEGLU (spatial_transformer.py:37-44): in (npix, 2F) = [x | gate] -> out (npix, F) = x * gelu(gate), the exact (erf) GELU
__global__ void k_geglu(const float *__restrict__ in, long npix, int F, float *__restrict__ out) {
    const long n = npix * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / F;
        const int f = (int)(i - pix * F);
        const float a = in[pix * 2 * F + f], g = in[pix * 2 * F + F + f];
        out[i] = a * (0.5f * g * (1.f + erff(g * 0.70710678118654752f)));
    }
}
// x (N, HW, C pitch) += v (N, C): the cross-attention over ONE context token (softmax over a single key is 1: the output is the same
// projected value vector at every query)
__global__ void k_add_rowvec(float *__restrict__ x, long pitch, long HW, long npix, int C, const float *__restrict__ v, long vpitch) {
    const long n = npix * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C;
        const int c = (int)(i - pix * C);
        x[pix * pitch + c] += v[(pix / HW) * vpitch + c];
    }
}

// ---- use_3d_aware=True (unet.py:566-570, 208-214, 613-614): the three planes of a tri-plane sit side by side, (B, C/3, H, 3W) ----
// (B, 3C, H, W) NCHW -> rolled NHWC (B, H, 3W, Cpad): pixel (y, p*W + x) channel c <- channel p*C + c; xs = x + x_cond likewise
__global__ void k_prep_inputs_3d(const float *__restrict__ x, const float *__restrict__ xc, int B, int C, int H, int W, int Cpad,
                                 float *__restrict__ xo, float *__restrict__ xs) {
    const long n = (long)B * H * 3 * W * Cpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long pix = i / Cpad;
        const int X = (int)(pix % (3 * W));
        const long by = pix / (3 * W);
        const int yy = (int)(by % H);
        const long b = by / H;
        const int pl = X / W, xx = X - pl * W;
        float v = 0.f, sm = 0.f;
        if (c < C) {
            const long src = ((b * 3 * C + pl * C + c) * H + yy) * W + xx;
            v = x[src];
            sm = xc ? v + xc[src] : v;
        }
        xo[i] = v;
        if (xs) xs[i] = sm;
    }
}
// rolled NCHW (B, C, H, 3W) -> (B, 3C, H, W)
__global__ void k_unroll_planes(const float *__restrict__ in, int B, int C, int H, int W, float *__restrict__ out) {
    const long n = (long)B * 3 * C * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        long r = i / W;
        const int yy = (int)(r % H); r /= H;
        const int ch = (int)(r % (3 * C));
        const long b = r / (3 * C);
        const int pl = ch / C, c = ch - pl * C;
        out[i] = in[((b * C + c) * H + yy) * (3L * W) + pl * W + xx];
    }
}
// per plane p: row sums over its W columns -> sums[b][p][y][c] (y < H) and column sums over the H rows -> sums[b][p][H + x][c]
__global__ __launch_bounds__(256) void k_plane_sums(const float *__restrict__ h, long pitch, int H, int W, int C, float *__restrict__ sums) {
    const int item = blockIdx.x % (H + W), bp = blockIdx.x / (H + W), pl = bp % 3, b = bp / 3;
    const float *base = h + ((long)b * H * 3 * W + (long)pl * W) * pitch;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f;
        if (item < H) for (int xx = 0; xx < W; ++xx) a += base[((long)item * 3 * W + xx) * pitch + c];
        else for (int yy = 0; yy < H; ++yy) a += base[((long)yy * 3 * W + (item - H)) * pitch + c];
        sums[((long)bp * (H + W) + item) * C + c] = a;
    }
}
// y (B, H, 3W, 3C) = silu(cat[hn, m1, m2]) with hn = A h + B and, per plane (unet.py:210-213):
//   plane 0: m1 = mean over W of plane 1 (a function of y), m2 = mean over H of plane 2 (of x)
//   plane 1: m1 = mean over W of plane 0 (y),               m2 = mean over W of plane 2 (y)
//   plane 2: m1 = mean over H of plane 0 (x),               m2 = mean over H of plane 1 (x)
// (means of the normalised tensor = the affine of the raw means)
__global__ void k_gn_apply_3d(const float *__restrict__ h, long pitch, int B, int H, int W, int C, const float *__restrict__ cA,
                              const float *__restrict__ cB, const float *__restrict__ sums, float *__restrict__ y) {
    const long n = (long)B * H * 3 * W * C;
    const float iw = 1.f / (float)W, ih = 1.f / (float)H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const int X = (int)(pix % (3 * W));
        const long by = pix / (3 * W);
        const int yy = (int)(by % H);
        const int b = (int)(by / H);
        const int pl = X / W, xx = X - pl * W;
        const float a = cA[(long)b * C + c], bb = cB[(long)b * C + c];
        auto rowm = [&](int q) { return sums[(((long)b * 3 + q) * (H + W) + yy) * C + c] * iw; };
        auto colm = [&](int q) { return sums[(((long)b * 3 + q) * (H + W) + H + xx) * C + c] * ih; };
        const float m1 = pl == 0 ? rowm(1) : (pl == 1 ? rowm(0) : colm(0));
        const float m2 = pl == 0 ? colm(2) : (pl == 1 ? rowm(2) : colm(1));
        float *o = y + pix * 3 * C + c;
        o[0] = silu_f(fmaf(h[pix * pitch + c], a, bb));
        o[C] = silu_f(fmaf(m1, a, bb));
        o[2 * C] = silu_f(fmaf(m2, a, bb));
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

size_t conv_splitk_ws_bytes() { return (size_t)64 << 20; }

size_t conv_packed_floats(int Cout, int Cin_pad, int ks) { return (size_t)round_up(Cout, 64) * Cin_pad * ks * ks; }

int conv_pack_weights(const float *w, int Cout, int Cin, int Cin_pad, int ks, float *packed, hipStream_t st, int tf) {
    HL_REQUIRE(w && packed && Cin_pad % 16 == 0 && Cin <= Cin_pad && (ks == 1 || ks == 3), "conv_pack_weights: bad argument");
    const int rows = round_up(Cout, 64);
    hipLaunchKernelGGL(k_pack_conv, dim3(1024), dim3(256), 0, st, w, Cout, Cin, Cin_pad, ks, rows, packed, tf);
    return check_launch("k_pack_conv");
}

size_t conv_packed_bf3_bytes(int Cout, int Cin_pad, int ks) {
    const int rows = round_up(Cout, 64);
    return rows % 96 == 0 ? (size_t)rows * Cin_pad * ks * ks * 6 : 0;
}

int conv_pack_weights_bf3(const float *w, int Cout, int Cin, int Cin_pad, int ks, void *packed, hipStream_t st, int tf) {
    HL_REQUIRE(w && packed && Cin_pad % 16 == 0 && Cin <= Cin_pad && (ks == 1 || ks == 3), "conv_pack_weights_bf3: bad argument");
    const int rows = round_up(Cout, 64);
    hipLaunchKernelGGL(k_pack_conv_bf3, dim3(1024), dim3(256), 0, st, w, Cout, Cin, Cin_pad, ks, rows, static_cast<unsigned short *>(packed), tf);
    return check_launch("k_pack_conv_bf3");
}

size_t conv_packed_wino_bytes(int Cout, int Cin_pad, int ks) {
    return (ks == 3 && Cout % 64 == 0 && Cin_pad % 8 == 0) ? (size_t)Cout * Cin_pad * 16 * sizeof(float) : 0;
}

int conv_pack_weights_wino(const float *w, int Cout, int Cin, int Cin_pad, float *packed, hipStream_t st, int tf) {
    HL_REQUIRE(w && packed && Cout % 64 == 0 && Cin_pad % 8 == 0 && Cin <= Cin_pad, "conv_pack_weights_wino: bad argument");
    hipLaunchKernelGGL(k_pack_conv_wino, dim3((unsigned)std::min<long>(8192, (long)(Cout >> 5) * (Cin_pad >> 3))), dim3(256), 0, st, w, Cout, Cin, Cin_pad, packed, tf);
    return check_launch("k_pack_conv_wino");
}

size_t conv_packed_wino4_bytes(int Cout, int Cin_pad, int ks) {   // (a ragged Cout - the 27-channel output convolution - is padded with zero rows to 32)
    return (ks == 3 && (Cout % 32 == 0 || Cout < 32) && Cin_pad % 8 == 0) ? (size_t)round_up(Cout, 32) * Cin_pad * 36 * sizeof(float) : 0;
}

int conv_pack_weights_wino4(const float *w, int Cout, int Cin, int Cin_pad, float *packed, hipStream_t st, int tf) {
    HL_REQUIRE(w && packed && (Cout % 32 == 0 || Cout < 32) && Cin_pad % 8 == 0 && Cin <= Cin_pad, "conv_pack_weights_wino4: bad argument");
    hipLaunchKernelGGL(k_pack_conv_wino4, dim3((unsigned)std::min<long>(8192, (long)(round_up(Cout, 32) >> 5) * (Cin_pad >> 3))), dim3(256), 0, st, w, Cout, Cin, Cin_pad, packed, tf);
    return check_launch("k_pack_conv_wino4");
}

static inline long hw_o_early(const ConvArgs &a) { return (long)a.out.H * a.out.W; }

// k_conv_h16 is taken from this many workgroups on.  The default (48; HL_H16_MIN_BLOCKS, read ONCE) is what the network dispatch uses; the
// unit tests that run the kernel on single tiles set it through hl_debug_set_h16_min_blocks (no getenv per convolution launch).
static long g_h16_min_blocks = -1;
static long h16_min_blocks() {
    static const long dflt = [] { const char *e = getenv("HL_H16_MIN_BLOCKS"); return e ? atol(e) : 48L; }();
    return g_h16_min_blocks >= 0 ? g_h16_min_blocks : dflt;
}
void set_h16_min_blocks(long v) { g_h16_min_blocks = v; }

int conv2d(const ConvArgs &a, hipStream_t st) {
    a.path = 0;
    HL_REQUIRE(a.in.p && a.w && a.out.p, "conv2d: null tensor");
    HL_REQUIRE(a.in.C % 16 == 0, "conv2d: Cin (%d) must be padded to a multiple of 16", a.in.C);
    HL_REQUIRE(a.ks == 1 || a.ks == 3, "conv2d: kernel size %d", a.ks);
    HL_REQUIRE(a.stride == 1 || (a.stride == 2 && !a.ups), "conv2d: stride/upsample combination");
    HL_REQUIRE((a.in.pitch % 4) == 0 && ((uintptr_t)a.in.p % 16) == 0, "conv2d: input must be 16-byte aligned per pixel");
    ConvK p{};
    p.in = a.in.p; p.in_pitch = a.in.pitch; p.N = a.in.N; p.Hin = a.in.H; p.Win = a.in.W; p.Cin = a.in.C;
    p.Hout = a.out.H; p.Wout = a.out.W; p.ks = a.ks; p.stride = a.stride; p.ups = a.ups; p.taps = a.ks * a.ks;
    const int pad = a.ks / 2;
    const int Hv = a.ups ? 2 * a.in.H : a.in.H, Wv = a.ups ? 2 * a.in.W : a.in.W;
    HL_REQUIRE(a.out.H == (Hv + 2 * pad - a.ks) / a.stride + 1 && a.out.W == (Wv + 2 * pad - a.ks) / a.stride + 1 &&
                   a.out.N == a.in.N, "conv2d: output shape mismatch");
    p.w = a.w; p.w_bf3 = a.w_bf3; p.w_wino = a.w_wino; p.Ktot = (long)a.in.C * p.taps; p.bias = a.bias; p.Cout = a.Cout; p.wrows = round_up(a.Cout, 64);
    p.cA = a.coefA; p.cB = a.coefB; p.act = a.act; p.gn = a.gn;
    p.st1_cg = a.st_cg > 0 ? a.st_cg : std::max(1, a.Cout / 32); p.st1_c0 = a.st_c0; p.st2_cg = a.st2_cg > 0 ? a.st2_cg : std::max(1, a.Cout / 32); p.st2_c0 = a.st2_c0;
    const bool gn_on = a.coefA != nullptr || a.gn.gt != nullptr;      // a GroupNorm affine in front of the convolution (arrays, or formed in the kernels)
    HL_REQUIRE(!gn_on || a.coefA || a.gn.C == a.in.C, "conv2d: GnSrc covers %d of %d channels", a.gn.C, a.in.C);
    p.out = a.out.p; p.out_pitch = a.out.pitch; p.res = a.res; p.res_pitch = a.res_pitch;
    p.out2 = a.out2; p.out2_pitch = a.out2_pitch; p.res2 = a.res2; p.res2_pitch = a.res2_pitch;
    p.out_nchw = a.out_nchw;
    p.M = (long)a.out.N * a.out.H * a.out.W;
    const long M = p.M;
    const int cpad = p.wrows;
    const int nk = (a.in.C / 16) * p.taps;
    // tile configs: 0 = 128x96 (4 waves of 32x96: 48 accumulators -> 4 waves/SIMD, 4 workgroups/CU = 1024 slots),
    // 1 = 128x32 (Cout <= 32), 2 = 64x64.  Small-M layers keep the efficient main tile and split K instead
    // (deterministic slabs + k_splitk_finish) until the grid covers the chip.
    const long main_blocks = ((M + 127) / 128) * (cpad / 96);
    static const int split_cap = [] { const char *e_ = getenv("HL_MAX_SPLITS"); return e_ ? atoi(e_) : 16; }();   // developer knob (read once)
    const int max_splits = a.splitk_ws ? (nk / 8 < split_cap ? nk / 8 : split_cap) : 1;
    int cfg;
    long blocks;
    if (cpad % 96 == 0 && main_blocks * (max_splits > 0 ? max_splits : 1) >= 192) { cfg = 0; blocks = main_blocks; }
    else if (a.Cout <= 32) { cfg = 1; blocks = (M + 127) / 128; }
    else { cfg = 2; blocks = ((M + 63) / 64) * (cpad / 64); }
    // main-tile layers that need no upsampling go through k_conv_dma (GroupNorm materialised by k_gn_apply first);
    // the 8-wave 256x96 tile when that covers at least half the chip (2 workgroups/CU = 512 slots), else 4 waves x 128x96
    constexpr int dma_thr = 256;      // 256x96 tiles when they give at least this many workgroups
    constexpr long wino_thr = 512;    // Winograd: workgroups wanted per launch (smaller layers split the input channels)
    constexpr long wino_min = 384;    // fewer even after splitting: direct kernel
    const bool dma = cfg == 0 && (!gn_on || a.act_ws) &&
                     (long)a.in.N * a.in.H * a.in.W * a.in.pitch * 4 < (1L << 31) && (long)cpad * p.Ktot * 4 < (1L << 31);
    const long blocks8 = ((M + 255) / 256) * (cpad / 96);
    const bool tile8 = dma && blocks8 >= dma_thr;
    if (tile8) blocks = blocks8;
    int splits = 1;
    const long target = cfg == 2 ? 1280 : (tile8 ? 512 : (dma ? 768 : 1024));
    if (a.splitk_ws && blocks <= target / 2 && nk >= 16) {
        splits = (int)(target / blocks);
        if (splits > nk / 8) splits = nk / 8;
        if (splits > split_cap) splits = split_cap;
        while (splits > 1 && (size_t)splits * M * a.Cout * sizeof(float) > a.splitk_ws_bytes) --splits;
        if (splits < 1) splits = 1;
    }
    p.kt_per = (nk + splits - 1) / splits;
    splits = (nk + p.kt_per - 1) / p.kt_per;
    p.partial = splits > 1 ? a.splitk_ws : nullptr;
    const int mode = gn_on ? (a.act ? 2 : 1) : 0;
    HL_REQUIRE(gn_on || !a.act, "conv2d: SiLU without the GroupNorm affine is not used by the UNet");
    // GroupNorm statistics of the output (ConvK::st1 / st2): slots of 32 consecutive pixels must not straddle images
    a.stat_slots = 0;
    const bool st_rows32 = a.stats && !a.out_nchw && hw_o_early(a) % 32 == 0;
    auto finish = [&](const char *what) -> int {   // split-K: the slab sum (+ statistics when wanted)
        int rc = check_launch(what);
        if (rc) return rc;
        static const int abl_ = [] { const char *e_ = getenv("HL_ABL_SKIP"); return e_ ? atoi(e_) : 0; }();   // TIMING ablation (wrong results)
        if (abl_ & 1) { if (st_rows32 && M % 32 == 0) a.stat_slots = (int)(hw_o_early(a) / 32); return HL_OK; }
        if (st_rows32 && M % 32 == 0) {
            p.st1 = a.stats; p.st2 = a.out2 ? a.stats2 : nullptr;
            a.stat_slots = (int)(hw_o_early(a) / 32);
            hipLaunchKernelGGL(k_splitk_finish_st, dim3((unsigned)(M / 32), (unsigned)((a.Cout + 63) / 64)), dim3(256), 0, st, p, splits);
        } else {
            long gf = (M * a.Cout + 255) / 256;
            if (gf > 2048) gf = 2048;
            hipLaunchKernelGGL(k_splitk_finish, dim3((unsigned)gf), dim3(256), 0, st, p, splits);
        }
        return check_launch("k_splitk_finish");
    };
    const long hw_o = (long)a.out.H * a.out.W;
#define HL_CONV_GO(WM_, WN_, MT_, NT_, GRID)                                                         \
    do {                                                                                             \
        constexpr int BM_ = WM_ * MT_ * 32, BN_ = WN_ * NT_ * 32;                                    \
        const int nimg = (int)((BM_ + hw_o - 1) / hw_o) + 1;                                         \
        const size_t shm = ((size_t)2 * (BM_ + BN_) * 20 + (mode ? (size_t)nimg * 2 * a.in.C + COEF_SCR_FLOATS : 0)) * sizeof(float); \
        HL_REQUIRE(shm <= 160 * 1024, "conv2d: LDS request %zu too large", shm);                     \
        if (a.ups) {   /* nearest x2 + conv: only ever follows a raw tensor (unet.py:77-79) */                \
            HL_REQUIRE(mode == 0, "conv2d: upsample with a GroupNorm prologue is not used by the UNet");    \
            hipLaunchKernelGGL((k_conv<WM_, WN_, MT_, NT_, 0, true>), GRID, dim3(256), shm, st, p);          \
        } else if (mode == 0) hipLaunchKernelGGL((k_conv<WM_, WN_, MT_, NT_, 0, false>), GRID, dim3(256), shm, st, p); \
        else if (mode == 1) hipLaunchKernelGGL((k_conv<WM_, WN_, MT_, NT_, 1, false>), GRID, dim3(256), shm, st, p);   \
        else hipLaunchKernelGGL((k_conv<WM_, WN_, MT_, NT_, 2, false>), GRID, dim3(256), shm, st, p);                  \
    } while (0)
    // 3x3 / stride-1 layers: Winograd F(2x2,3x3) with 16x8-pixel x 64-channel workgroups; layers that do not fill the chip
    // that way split the input channels into slabs (the output transform is linear: k_splitk_finish sums outputs)
    const long wino_blocks = (long)a.out.N * (a.out.H / 8) * (a.out.W / 16) * (a.Cout / 64);
    bool wino = dma && a.w_wino && !a.w_bf3 && a.ks == 3 && a.stride == 1 && a.out.H % 8 == 0 && a.out.W % 16 == 0 &&
                a.Cout % 64 == 0 && (long)a.Cout * a.in.C * 64 < (1L << 31);   // (stride 1: out = in, or 2x in when upsampling)
    int wsplits = 1;
    if (wino && wino_blocks < wino_thr) {
        const int nkt8 = a.in.C / 8;
        // the most slabs that still fit ONE round of workgroups (2 per CU): rounding up instead put 576 workgroups on 512 slots - a second,
        // nearly empty round (measured: 66 -> 63 us at 32x32, 67 -> 57 us at 16x16, 122 -> 102 us with 1536 input channels)
        wsplits = (int)(wino_thr / wino_blocks);
        if (wsplits < 2) wsplits = 2;
        if (wsplits > nkt8 / 6) wsplits = nkt8 / 6;              // at least 6 k-tiles (48 channels) per slab
        if (wsplits > 16) wsplits = 16;
        while (wsplits > 1 && (size_t)wsplits * M * a.Cout * sizeof(float) > a.splitk_ws_bytes) --wsplits;
        if (!a.splitk_ws || wsplits < 2 || wino_blocks * wsplits < wino_min) wino = false;
    }
    if (wino) {
        splits = wsplits;
        p.kt_per = (a.in.C / 8 + splits - 1) / splits;
        splits = (a.in.C / 8 + p.kt_per - 1) / p.kt_per;
        p.partial = splits > 1 ? a.splitk_ws : nullptr;
    }
    // the same layers by Winograd F(4x4,3x3) (32x16-pixel x 32-channel workgroups, a quarter of the direct multiplies) where that
    // alone fills the chip
    constexpr long wino4_thr = 512;
    // (the 27-channel NCHW output convolution takes the F(4x4) kernel too: its weights are padded to 32 rows, the epilogue stores 27)
    const bool small_nchw = a.out_nchw && a.Cout < 32 && !a.res && !a.out2 && !a.stats && (!gn_on || a.act_ws) &&
                            (long)a.in.N * a.in.H * a.in.W * a.in.pitch * 4 < (1L << 31);
    const long wino4_blocks = (long)a.out.N * (a.out.H / 16) * (a.out.W / 32) * ((a.Cout + 31) / 32);
    const bool wino4_ok = (dma || small_nchw) && a.w_wino4 && !a.w_bf3 && a.ks == 3 && a.stride == 1 && a.out.H % 16 == 0 && a.out.W % 32 == 0 &&
                          (a.Cout % 32 == 0 || small_nchw) && (long)round_up(a.Cout, 32) * a.in.C * 144 < (1L << 31);
    bool wino4 = wino4_ok && wino4_blocks >= wino4_thr;
    // k_conv_wino4w: the same arithmetic with 64 output channels per workgroup at ONE workgroup per CU (hl_conv_wino4w.hip).  With W
    // = 32x16-pixel x 64-channel workgroups it is taken
    //   * from three rounds of the 256 CUs on (W >= 768: the 256-pixel level at batch >= 2),
    //   * where one round nearly fills the chip (160 <= W <= 256: the 64-pixel level at batch 4 - 192 workgroups run 140 us where the
    //     768 smaller workgroups of the F(2x2) kernel take 188 us),
    //   * below that with the input channels split into slabs until W x slabs reaches one round (k_splitk_finish sums them).
    // In between (1.5 rounds: 128x128 at batch 4, 384 workgroups, 151 us against 131 us) the two-workgroups-per-CU kernels keep the layer.
    static const int w4w_mode = [] { const char *e_ = getenv("HL_WINO4W"); return e_ ? atoi(e_) : 1; }();   // developer switch, read once: 0 = off, 2 = wherever it can run
    const long w4w_blocks = wino4_blocks / 2;
    bool wino4w = false;
    int w4w_splits = 1;
    if (wino4_ok && w4w_mode != 0 && a.Cout % 64 == 0 && w4w_blocks > 0) {
        const int nkt8 = a.in.C / 8;
        if (w4w_blocks >= 768 || (w4w_blocks >= 160 && w4w_blocks <= 256) || (w4w_mode == 2 && w4w_blocks >= 160)) wino4w = true;
        else if (w4w_blocks < 160 && a.splitk_ws) {
            w4w_splits = (int)(256 / w4w_blocks);
            if (w4w_splits > nkt8 / 8) w4w_splits = nkt8 / 8;            // at least 8 k-tiles (64 channels) per slab
            while (w4w_splits > 1 && (size_t)w4w_splits * M * a.Cout * sizeof(float) > a.splitk_ws_bytes) --w4w_splits;
            wino4w = w4w_splits >= 2 && w4w_blocks * w4w_splits >= 128;
            if (!wino4w) w4w_splits = 1;
        }
    }
    if (wino4w) wino4 = true;
    if (wino4) {
        wino = false;
        splits = wino4w ? w4w_splits : 1;
        p.kt_per = (a.in.C / 8 + splits - 1) / splits;
        splits = (a.in.C / 8 + p.kt_per - 1) / p.kt_per;
        p.partial = splits > 1 ? a.splitk_ws : nullptr;
    }
    // 16-bit operands (opt-in modes): the 3x3 / stride-1 layers k_conv_h16 covers; the rest of the mode stays on k_conv_bf3 / fp32
    // 16-bit operands (opt-in modes): the 3x3 / 1x1 stride-1 layers k_conv_h16 / k_conv1_h16 cover, from h16_min_blocks() workgroups on; a 3x3
    // layer with fewer tiles splits its input channels into slabs of >= 2 chunks (k_splitk_finish sums them) until ~128 workgroups run
    const long h16_blocks = ((long)a.out.N * a.out.H * a.out.W / 256) * (a.Cout / 192);
    bool h16 = a.w_h16 && (!gn_on || (a.act_ws && !a.ups)) && !a.out_nchw &&
               conv_h16_applies(a.out.H, a.out.W, a.in.C, a.Cout, a.ks, a.stride, a.ups) &&
               (long)a.in.N * a.in.H * a.in.W * a.in.pitch * 4 < (1L << 31) && a.in.pitch % 4 == 0 && h16_blocks > 0;
    int h16_splits = 1;
    if (h16 && h16_blocks < h16_min_blocks()) {
        const int nch32 = a.in.C / 32;
        if (a.ks == 3 && a.splitk_ws && !a.out2 && h16_blocks >= 4) {
            h16_splits = (int)std::min<long>(std::min<long>(128 / h16_blocks, nch32 / 2), 16);
            while (h16_splits > 1 && (size_t)h16_splits * M * a.Cout * sizeof(float) > a.splitk_ws_bytes) --h16_splits;
        }
        if (h16_splits < 2 || h16_blocks * h16_splits < h16_min_blocks()) { h16 = false; h16_splits = 1; }
    }
    // 1x1 / stride-1 layers of the DEFAULT mode on the 16-bit matrix pipe with fp16x2 products (k_conv1_h2): once off the fp32 pipe they are bound by HBM, the
    // fp32 kernel takes twice as long.  From h2_min_blocks workgroups of 256 pixels x 192 channels on (fewer: the split-K fp32 path keeps the layer).
    static const long h2_min_blocks = [] { const char *e_ = getenv("HL_H2_MIN_BLOCKS"); return e_ ? atol(e_) : 12L; }();   // developer knob (read once); < 0 disables (12: with the 128-pixel tiles; 48 with the 256-pixel ones)
    const long h2_blocks = (M / 256) * (a.Cout / 192);
    const bool h2 = !h16 && a.w_h2 && h2_min_blocks >= 0 && (!gn_on || a.act_ws) && !a.out_nchw && !a.w_bf3 &&
                    conv1_h2_applies(a.out.H, a.out.W, a.in.C, a.Cout, a.ks, a.stride, a.ups) &&
                    (long)a.in.N * a.in.H * a.in.W * a.in.pitch * 4 < (1L << 31) && a.in.pitch % 4 == 0 && h2_blocks >= h2_min_blocks;
    // 3x3 / stride-1 layers of the default mode in the same arithmetic, from 100 workgroups of 256 pixels x 192 channels on: k_conv_h2s, a direct convolution on
    // 8x16-pixel tiles with TWO workgroups per CU (late round 5).  Same box, forward wall time: B = 1 12.8 -> 12.2 ms, B = 4 32.5 -> 30.4, B = 8 58.9 -> 54.9 (both
    // small-tile kernels).  The first version of the kernel (16x16 tiles, one workgroup per CU: k_conv_h16<., 2>, HL_H2_SMALL=0) won only where a layer was about one
    // round of workgroups: alone it beat k_conv_wino4w by 12 % on the 256-pixel level and the forward's wall time did not move - a kernel that owns whole CUs cannot fill
    // the other encoder tower's bubbles, and nothing overlapped its own prologue / staging / epilogue (profiles/r05_unet_fill_experiments.md, sections 6 - 8).
    static const long h3_min_blocks = [] { const char *e_ = getenv("HL_H2_CONV3_MIN_BLOCKS"); return e_ ? atol(e_) : 100L; }();   // developer knobs (read once); min < 0 disables
    static const long h3_max_blocks = [] { const char *e_ = getenv("HL_H2_CONV3_MAX_BLOCKS"); return e_ ? atol(e_) : (1L << 40); }();
    const bool h3_base = !h16 && !h2 && a.w_h2 && a.ks == 3 && h3_min_blocks >= 0 && (!gn_on || (a.act_ws && !a.ups)) && !a.out_nchw && !a.w_bf3 &&
                         conv_h16_applies(a.out.H, a.out.W, a.in.C, a.Cout, a.ks, a.stride, a.ups) &&
                         (long)a.in.N * a.in.H * a.in.W * a.in.pitch * 4 < (1L << 31) && a.in.pitch % 4 == 0;
    bool h3 = h3_base && h16_blocks >= h3_min_blocks && h16_blocks <= h3_max_blocks;
    // below that: the input channels split into slabs of >= 2 chunks until about one round of 128-pixel workgroups runs (k_splitk_finish[_st] sums the slabs)
    static const long h3_split_min = [] { const char *e_ = getenv("HL_H2_CONV3_SPLIT_MIN"); return e_ ? atol(e_) : 8L; }();   // developer knob (read once); < 0: no split-K on this kernel
    int h3_splits = 1;
    if (h3_base && !h3 && h3_split_min >= 0 && h16_blocks >= h3_split_min && h16_blocks < h3_min_blocks && a.splitk_ws && !a.out2) {
        static const long h3_min_chunks = [] { const char *e_ = getenv("HL_H2_SPLIT_MIN_CHUNKS"); return e_ ? std::max(1L, atol(e_)) : 2L; }();   // developer knob (read once): chunks of 32 input channels per slab, at least
        static const long h3_max_splits = [] { const char *e_ = getenv("HL_H2_SPLIT_MAX"); return e_ ? std::max(1L, atol(e_)) : 16L; }();
        static const long h3_target = [] { const char *e_ = getenv("HL_H2_SPLIT_TARGET"); return e_ ? std::max(1L, atol(e_)) : 256L; }();                // workgroups (of 256 pixels x 192 channels) aimed at
        h3_splits = (int)std::min<long>(std::min<long>(h3_target / h16_blocks, (a.in.C / 32) / h3_min_chunks), h3_max_splits);
        while (h3_splits > 1 && (size_t)h3_splits * M * a.Cout * sizeof(float) > a.splitk_ws_bytes) --h3_splits;
        if (h3_splits >= 2) h3 = true; else h3_splits = 1;
    }
    // 3x3 / stride-2 layers (Downsample) of the default mode in the same arithmetic (k_conv_h2d), from h3d_min_blocks workgroups' worth of output (256 pixels x 192 channels) on
    static const long h3d_min_blocks = [] { const char *e_ = getenv("HL_H2_CONV3S2_MIN_BLOCKS"); return e_ ? atol(e_) : 32L; }();   // developer knob (read once); < 0 disables
    const bool h3d = !h16 && !h2 && !h3 && a.w_h2 && a.ks == 3 && a.stride == 2 && !a.ups && mode == 0 && h3d_min_blocks >= 0 && !a.out_nchw && !a.w_bf3 &&
                     conv3_h2d_applies(a.out.H, a.out.W, a.in.C, a.Cout) && (long)a.in.N * a.in.H * a.in.W * a.in.pitch * 4 < (1L << 31) && a.in.pitch % 4 == 0 &&
                     h16_blocks >= h3d_min_blocks;
    if (a.plan_only) {   // which weight layout will this launch read?  (single-op entry points pack only that one)
        a.path = h16 ? 5 : (h2 || h3 || h3d) ? 6 : (wino4 ? 3 : ((dma && wino) ? 1 : ((dma && a.w_bf3 && (long)cpad * p.Ktot * 6 < (1L << 31)) ? 2 : 0)));
        return HL_OK;
    }
    if (h16) {
        a.path = 5;
        if (mode != 0) {   // GroupNorm(+SiLU) materialised once, as the 16-bit image the kernel stages without conversion
            HL_REQUIRE((size_t)a.in.pixels() * a.in.C * sizeof(float) <= a.act_ws_bytes, "conv2d: act scratch too small");
            const long npix = a.in.pixels();
            long g = (npix * (a.in.C / 4) + 255) / 256;
            if (g > 4096) g = 4096;
            if (a.coefA == nullptr) {
                const int ppw = gn_gs_ppw(a.in.H * a.in.W, a.in.N, a.in.C);
                hipLaunchKernelGGL(k_gn_apply_gs<1>, dim3((unsigned)((a.in.H * a.in.W + ppw - 1) / ppw), (unsigned)a.in.N), dim3(256), (size_t)(2 * a.in.C + COEF_SCR_FLOATS) * sizeof(float), st,
                                   a.in.p, a.in.pitch, a.in.H * a.in.W, a.in.C, a.gn, a.in.N, a.act, (void *)a.act_ws, a.h16_fp16, ppw);
            } else
            hipLaunchKernelGGL(k_gn_apply_h16, dim3((unsigned)g), dim3(256), 0, st, a.in.p, a.in.pitch, (long)a.in.H * a.in.W, npix, a.in.C, a.coefA,
                               a.coefB, a.act, reinterpret_cast<unsigned short *>(a.act_ws), a.h16_fp16);
            p.in = a.act_ws; p.in_pitch = a.in.C; p.cA = nullptr; p.cB = nullptr; p.act = 0; p.gn = GnSrc{};
            p.in16 = 1;                                  // in_pitch counts 16-bit elements now
            if (a.ev_mid) { hipEventRecord(a.ev_mid, st); a.ev_mid_used = 1; }
        }
        p.w_bf3 = a.w_h16;
        splits = h16_splits;
        p.kt_per = (a.in.C / 32 + splits - 1) / splits;                  // chunks of 32 input channels per slab
        splits = (a.in.C / 32 + p.kt_per - 1) / p.kt_per;
        p.partial = splits > 1 ? a.splitk_ws : nullptr;
        p.n_nblocks = a.Cout / 192;
        p.n_mtiles = (int)((long)a.out.N * a.out.H * a.out.W / 256);   // 16x16-pixel tiles (3x3) / runs of 256 pixels (1x1)
        if (a.stats) {   // statistics from the epilogue: slot = (tile, round) = 128 pixels
            p.st1 = a.stats; p.st2 = a.out2 ? a.stats2 : nullptr;
            a.stat_slots = a.out.H * a.out.W / 128;
        }
        if (splits > 1) {   // the finish kernel emits the statistics
            p.st1 = p.st2 = nullptr;
            a.stat_slots = 0;
            int rc = conv_h16_launch(p, a.h16_fp16, st, splits);
            if (rc) return rc;
            return finish("k_conv_h16");
        }
        return conv_h16_launch(p, a.h16_fp16, st);
    }
    if (h3d) {
        a.path = 6;
        p.in16 = 0; p.w_bf3 = a.w_h2; p.partial = nullptr;
        p.wsc = conv_h2_wscale(a.w_h2, a.Cout, a.in.C, a.ks);
        p.xs_gt = a.in_stats; p.xs_hw = a.in.H * a.in.W; p.xs_max = a.in_absmax;
        p.n_nblocks = a.Cout / 192;
        p.n_mtiles = (int)((long)a.out.N * a.out.H * a.out.W / 128);
        if (a.stats) {
            p.st1 = a.stats; p.st2 = a.out2 ? a.stats2 : nullptr;
            a.stat_slots = a.out.H * a.out.W / 128;
        }
        return conv3_h2d_launch(p, st);
    }
    if (h3) {
        a.path = 6;
        p.in16 = 0;
        // GroupNorm(+SiLU) of the input: applied by k_conv_h2s while it stages the patch (its two workgroups per CU hide the VALU) - no pass over the tensor
        static const int h3_fuse = [] { const char *e_ = getenv("HL_H2_FUSE_GN"); return e_ ? atoi(e_) : 1; }();   // developer knob (read once)
        static const int h3_small_ = [] { const char *e_ = getenv("HL_H2_SMALL"); return e_ ? atoi(e_) : 1; }();      // 0: the 16x16-pixel kernel, one workgroup per CU
        const bool fuse = mode != 0 && h3_fuse && h3_small_ && !a.ups && a.in.C <= 4096;
        if (mode != 0 && !fuse) {   // GroupNorm(+SiLU) materialised once, as the two-plane image the kernel stages without conversion (the same bytes as fp32)
            HL_REQUIRE((size_t)a.in.pixels() * a.in.C * sizeof(float) <= a.act_ws_bytes, "conv2d: act scratch too small");
            const long npix = a.in.pixels();
            if (a.coefA == nullptr) {
                const int ppw = gn_gs_ppw(a.in.H * a.in.W, a.in.N, a.in.C);
                hipLaunchKernelGGL(k_gn_apply_gs<1>, dim3((unsigned)((a.in.H * a.in.W + ppw - 1) / ppw), (unsigned)a.in.N), dim3(256), (size_t)(2 * a.in.C + COEF_SCR_FLOATS) * sizeof(float), st,
                                   a.in.p, a.in.pitch, a.in.H * a.in.W, a.in.C, a.gn, a.in.N, a.act, (void *)a.act_ws, 2, ppw);
            } else {
                long g = (npix * (a.in.C / 4) + 255) / 256;
                if (g > 4096) g = 4096;
                hipLaunchKernelGGL(k_gn_apply_h16, dim3((unsigned)g), dim3(256), 0, st, a.in.p, a.in.pitch, (long)a.in.H * a.in.W, npix, a.in.C, a.coefA,
                                   a.coefB, a.act, reinterpret_cast<unsigned short *>(a.act_ws), 2);
            }
            p.in = a.act_ws; p.in_pitch = a.in.C; p.cA = nullptr; p.cB = nullptr; p.act = 0; p.gn = GnSrc{};
            p.in16 = 2;
            if (a.ev_mid) { hipEventRecord(a.ev_mid, st); a.ev_mid_used = 1; }
        }
        p.w_bf3 = a.w_h2;
        p.wsc = conv_h2_wscale(a.w_h2, a.Cout, a.in.C, a.ks);
        if (mode == 0) { p.xs_gt = a.in_stats; p.xs_hw = a.in.H * a.in.W; p.xs_max = a.in_absmax; }      // (raw input; a fused GroupNorm bounds its own output)
        splits = h3_splits;
        p.kt_per = (a.in.C / 32 + splits - 1) / splits;                  // chunks of 32 input channels per slab
        splits = (a.in.C / 32 + p.kt_per - 1) / p.kt_per;
        p.partial = splits > 1 ? a.splitk_ws : nullptr;
        p.n_nblocks = a.Cout / 192;
        p.n_mtiles = (int)((long)a.out.N * a.out.H * a.out.W / 256);
        if (a.stats) {
            p.st1 = a.stats; p.st2 = a.out2 ? a.stats2 : nullptr;
            a.stat_slots = a.out.H * a.out.W / 128;
        }
        if (splits > 1) {   // the finish kernel adds bias / residual and emits the statistics
            p.st1 = p.st2 = nullptr;
            a.stat_slots = 0;
            p.n_mtiles *= 2;
            int rc = conv3_h2s_launch(p, st, splits);
            if (rc) return rc;
            return finish("k_conv_h2s");
        }
        if (h3_small_) {   // 8x16-pixel tiles, two workgroups per CU
            p.n_mtiles *= 2;
            return conv3_h2s_launch(p, st);
        }
        return conv3_h2_launch(p, st);
    }
    if (h2) {
        a.path = 6;
        static const int h2_fuse = [] { const char *e_ = getenv("HL_H2_FUSE_GN"); return e_ ? atoi(e_) : 1; }();     // developer knobs (read once)
        static const int h2_small_ = [] { const char *e_ = getenv("HL_H2_SMALL1"); return e_ ? atoi(e_) : 1; }();
        const bool fuse1 = mode != 0 && h2_fuse && h2_small_ && a.in.C <= 4096;      // GroupNorm(+SiLU) applied by k_conv1_h2s while staging
        if (mode != 0 && !fuse1) {   // GroupNorm(+SiLU) materialised once as dense fp32 (the kernel splits into its two fp16 planes while staging)
            HL_REQUIRE((size_t)a.in.pixels() * a.in.C * sizeof(float) <= a.act_ws_bytes, "conv2d: act scratch too small");
            const long npix = a.in.pixels();
            if (a.coefA == nullptr) {
                const int ppw = gn_gs_ppw(a.in.H * a.in.W, a.in.N, a.in.C);
                hipLaunchKernelGGL(k_gn_apply_gs<0>, dim3((unsigned)((a.in.H * a.in.W + ppw - 1) / ppw), (unsigned)a.in.N), dim3(256), (size_t)(2 * a.in.C + COEF_SCR_FLOATS) * sizeof(float), st,
                                   a.in.p, a.in.pitch, a.in.H * a.in.W, a.in.C, a.gn, a.in.N, a.act, (void *)a.act_ws, 0, ppw);
            } else {
                long g = (npix * (a.in.C / 4) + 255) / 256;
                if (g > 4096) g = 4096;
                hipLaunchKernelGGL(k_gn_apply, dim3((unsigned)g), dim3(256), 0, st, a.in.p, a.in.pitch, (long)a.in.H * a.in.W, npix, a.in.C, a.coefA, a.coefB, a.act, a.act_ws);
            }
            p.in = a.act_ws; p.in_pitch = a.in.C; p.cA = nullptr; p.cB = nullptr; p.act = 0; p.gn = GnSrc{};
            if (a.ev_mid) { hipEventRecord(a.ev_mid, st); a.ev_mid_used = 1; }
        }
        p.w_bf3 = a.w_h2; p.in16 = 0; p.partial = nullptr;
        p.wsc = conv_h2_wscale(a.w_h2, a.Cout, a.in.C, a.ks);
        if (mode == 0) { p.xs_gt = a.in_stats; p.xs_hw = a.in.H * a.in.W; p.xs_max = a.in_absmax; }
        p.n_nblocks = a.Cout / 192;
        p.n_mtiles = (int)(M / 256);
        if (a.stats) {   // statistics from the epilogue (128 pixels of one image per round)
            p.st1 = a.stats; p.st2 = a.out2 ? a.stats2 : nullptr;
            a.stat_slots = 1;
        }
        if (h2_small_) {   // 128-pixel tiles, two workgroups per CU
            p.n_mtiles *= 2;
            p.kt_per = a.in.C / 48;
            // few workgroups (the 16- and 32-pixel levels, batch 1): the input channels split into slabs of >= 4 chunks until ~256 workgroups run (k_splitk_finish[_st] sums them)
            static const long h2_split_max = [] { const char *e_ = getenv("HL_H2_SPLIT_MAX_BLOCKS"); return e_ ? atol(e_) : 64L; }();   // developer knob (read once); 0: no split-K
            if (h2_blocks < h2_split_max && a.splitk_ws && !a.out2) {
                int sp = (int)std::min<long>(std::min<long>(128 / std::max<long>(h2_blocks, 1), (a.in.C / 48) / 4), 8);
                while (sp > 1 && (size_t)sp * M * a.Cout * sizeof(float) > a.splitk_ws_bytes) --sp;
                if (sp >= 2) {
                    p.kt_per = (a.in.C / 48 + sp - 1) / sp;
                    splits = (a.in.C / 48 + p.kt_per - 1) / p.kt_per;
                    p.partial = a.splitk_ws;
                    p.st1 = p.st2 = nullptr;
                    a.stat_slots = 0;
                    int rc = conv1_h2s_launch(p, st, splits);
                    if (rc) return rc;
                    return finish("k_conv1_h2s");
                }
            }
            return conv1_h2s_launch(p, st);
        }
        return conv1_h2_launch(p, st);
    }
    bool blk4 = false;
    if (dma || wino4) {
        HL_REQUIRE(mode == 0 || !a.ups, "conv2d: upsample with a GroupNorm prologue is not used by the UNet");
        if (mode != 0) {   // materialise GroupNorm(+SiLU) once, then the DMA kernels read it raw
            HL_REQUIRE((size_t)a.in.pixels() * a.in.C * sizeof(float) <= a.act_ws_bytes, "conv2d: act scratch too small");
            const long npix = a.in.pixels();
            long g = (npix * (a.in.C / 4) + 255) / 256;
            if (g > 4096) g = 4096;
            if (wino4 && npix % 64 == 0 && (a.coefA || ((long)a.in.H * a.in.W) % 8 == 0)) {
                // for the F(4x4) kernel the normalised copy is channel-blocked, [C/8][pixel][8]: its patch DMA then reads 128 contiguous
                // bytes per four pixels instead of 32 per pixel (the gather rate of the LDS-DMA path is set by the number of distinct
                // segments: 33 B/ns/CU at 32 bytes, 148 at 128 - scripts/microbench/dma_bw.hip)
                blk4 = true;
                const int tp = 8;   // (8 pixels = 256 contiguous bytes per plane; larger tiles cost occupancy: 64 pixels 93 us, 8 pixels 70 us = the plain pass)
                const int pad = ((4 - (a.in.C / 4) % 16 + 16) % 16) * 4;   // row length / 4 = 4 (mod 16): the read-back groups (8 pixels x 2 halves of two planes) hit 16 different slots
                const size_t shb = (size_t)(tp * (a.in.C + pad) + (a.coefA ? 0 : 2 * a.in.C + COEF_SCR_FLOATS)) * sizeof(float);
                // coefficients formed in the kernel: a workgroup amortises them over `reps` tiles of its image (at most 16, at least ~2048 workgroups)
                int reps = 1;
                if (!a.coefA) {
                    const long tiles_img = (long)a.in.H * a.in.W / tp;
                    while (reps < 16 && tiles_img % (reps * 2) == 0 && npix / tp / (reps * 2) >= 2048) reps *= 2;
                }
                hipLaunchKernelGGL(k_gn_apply_blk, dim3((unsigned)(npix / tp / reps)), dim3(256), shb, st, a.in.p, a.in.pitch, (long)a.in.H * a.in.W, npix,
                                   a.in.C, a.coefA, a.coefB, a.act, a.act_ws, tp, a.in.C + pad, a.gn, a.in.N, reps);
            } else if (a.coefA == nullptr) {
                const int ppw = gn_gs_ppw(a.in.H * a.in.W, a.in.N, a.in.C);
                hipLaunchKernelGGL(k_gn_apply_gs<0>, dim3((unsigned)((a.in.H * a.in.W + ppw - 1) / ppw), (unsigned)a.in.N), dim3(256), (size_t)(2 * a.in.C + COEF_SCR_FLOATS) * sizeof(float), st,
                                   a.in.p, a.in.pitch, a.in.H * a.in.W, a.in.C, a.gn, a.in.N, a.act, (void *)a.act_ws, 0, ppw);
            } else
            hipLaunchKernelGGL(k_gn_apply, dim3((unsigned)g), dim3(256), 0, st, a.in.p, a.in.pitch, (long)a.in.H * a.in.W, npix,
                               a.in.C, a.coefA, a.coefB, a.act, a.act_ws);
            p.in = a.act_ws; p.in_pitch = a.in.C; p.cA = nullptr; p.cB = nullptr; p.act = 0; p.gn = GnSrc{};
            if (a.ev_mid) { hipEventRecord(a.ev_mid, st); a.ev_mid_used = 1; }
        }
        if (wino4 && wino4w) {
            // 64 output channels per workgroup, one wave per SIMD, accumulators in the accumulator registers (hl_conv_wino4w.hip)
            a.path = 3;
            p.w_wino = a.w_wino4;
            p.n_nblocks = a.Cout / 64;
            p.n_mtiles = a.out.N * (a.out.H / 16) * (a.out.W / 32);
            if (splits == 1 && a.stats && !a.out_nchw) {   // statistics from the epilogue: slot = (32x16 block, round, wave) = 64 pixels
                p.st1 = a.stats; p.st2 = a.out2 ? a.stats2 : nullptr;
                a.stat_slots = (a.out.H / 16) * (a.out.W / 32) * 8;
            }
            int rc = conv_wino4w_launch(p, a.ups, blk4 ? 1 : 0, splits, st);
            if (rc) return rc;
            if (splits > 1) return finish("k_conv_wino4w");
            return HL_OK;
        }
        if (wino4) {
            a.path = 3;
            p.w_wino = a.w_wino4;
            p.n_nblocks = (a.Cout + 31) / 32;
            p.n_mtiles = a.out.N * (a.out.H / 16) * (a.out.W / 32);
            const dim3 nblk((unsigned)(p.n_mtiles * p.n_nblocks), 1, splits);
            const size_t sh4 = (size_t)(36 * 256 + 2 * 1296 * 4) * sizeof(float);   // 76.5 KB: two workgroups per CU
            if (splits == 1 && a.stats && !a.out_nchw) {   // statistics from the epilogue: slot = (32x16 block, round, wave) = 64 pixels
                p.st1 = a.stats; p.st2 = a.out2 ? a.stats2 : nullptr;
                a.stat_slots = (a.out.H / 16) * (a.out.W / 32) * 8;
            }
            if (a.ups) hipLaunchKernelGGL((k_conv_wino4<true, false>), nblk, dim3(256), sh4, st, p);
            else if (blk4) hipLaunchKernelGGL((k_conv_wino4<false, true>), nblk, dim3(256), sh4, st, p);
            else hipLaunchKernelGGL((k_conv_wino4<false, false>), nblk, dim3(256), sh4, st, p);
            if (splits > 1) return finish("k_conv_wino4");
            return check_launch("k_conv_wino4");
        }
        if (wino) {
            a.path = 1;
            p.n_nblocks = a.Cout / 64;
            p.n_mtiles = a.out.N * (a.out.H / 8) * (a.out.W / 16);
            const dim3 nblk((unsigned)(p.n_mtiles * p.n_nblocks), 1, splits);
            const size_t sh1 = (size_t)2 * (8192 + 7 * 256) * sizeof(float);
            if (splits == 1 && a.stats && !a.out_nchw) {   // statistics from the epilogue: slot = (16x8 block, column parity)
                p.st1 = a.stats; p.st2 = a.out2 ? a.stats2 : nullptr;
                a.stat_slots = (a.out.H / 8) * (a.out.W / 16) * 2;
            }
            if (a.ups) hipLaunchKernelGGL((k_conv_wino<true>), nblk, dim3(256), sh1, st, p);
            else hipLaunchKernelGGL((k_conv_wino<false>), nblk, dim3(256), sh1, st, p);
            if (splits > 1) return finish("k_conv_wino");
            return check_launch("k_conv_wino");
        }
        p.n_nblocks = cpad / 96;
        p.n_mtiles = (int)((M + (tile8 ? 255 : 127)) / (tile8 ? 256 : 128));
        dim3 grid((unsigned)(p.n_mtiles * p.n_nblocks), 1, splits);
        const size_t shm8 = (size_t)3 * 352 * 16 * sizeof(float), shm4 = (size_t)3 * 224 * 16 * sizeof(float);
        if (splits == 1 && st_rows32 && !(a.w_bf3 && (long)cpad * p.Ktot * 6 < (1L << 31))) {   // statistics from the epilogue: slot = a wave's 32 rows
            p.st1 = a.stats; p.st2 = a.out2 ? a.stats2 : nullptr;
            a.stat_slots = (int)(hw_o_early(a) / 32);
        }
        if (a.w_bf3 && (long)cpad * p.Ktot * 6 < (1L << 31)) {   // fp32 emulated on the bf16 matrix pipe (opt-in)
            a.path = 2;
            const size_t s8 = (size_t)3 * (256 * 16 + 6 * 96 * 4) * sizeof(float), s4 = (size_t)3 * (128 * 16 + 6 * 96 * 4) * sizeof(float);
            if (a.bf16_single) {   // HL_CONV_BF16: bf16 activations x 16-bit weights
                if (tile8 && a.ups) hipLaunchKernelGGL((k_conv_bf3<8, true, 1>), grid, dim3(512), s8, st, p);
                else if (tile8) hipLaunchKernelGGL((k_conv_bf3<8, false, 1>), grid, dim3(512), s8, st, p);
                else if (a.ups) hipLaunchKernelGGL((k_conv_bf3<4, true, 1>), grid, dim3(256), s4, st, p);
                else hipLaunchKernelGGL((k_conv_bf3<4, false, 1>), grid, dim3(256), s4, st, p);
            } else if (tile8 && a.ups) hipLaunchKernelGGL((k_conv_bf3<8, true>), grid, dim3(512), s8, st, p);
            else if (tile8) hipLaunchKernelGGL((k_conv_bf3<8, false>), grid, dim3(512), s8, st, p);
            else if (a.ups) hipLaunchKernelGGL((k_conv_bf3<4, true>), grid, dim3(256), s4, st, p);
            else hipLaunchKernelGGL((k_conv_bf3<4, false>), grid, dim3(256), s4, st, p);
        } else
        if (tile8 && a.ups) hipLaunchKernelGGL((k_conv_dma<8, 3, true>), grid, dim3(512), shm8, st, p);
        else if (tile8) hipLaunchKernelGGL((k_conv_dma<8, 3, false>), grid, dim3(512), shm8, st, p);
        else if (a.ups) hipLaunchKernelGGL((k_conv_dma<4, 3, true>), grid, dim3(256), shm4, st, p);
        else hipLaunchKernelGGL((k_conv_dma<4, 3, false>), grid, dim3(256), shm4, st, p);
    } else if (cfg == 0) {
        p.n_mtiles = (int)((M + 127) / 128); p.n_nblocks = cpad / 96;
        dim3 grid((unsigned)(p.n_mtiles * p.n_nblocks), 1, splits);
        HL_CONV_GO(4, 1, 1, 3, grid);
    } else if (cfg == 1) {
        p.n_mtiles = (int)((M + 127) / 128); p.n_nblocks = 1;
        dim3 grid((unsigned)p.n_mtiles, 1, splits);
        HL_CONV_GO(4, 1, 1, 1, grid);
    } else {
        p.n_mtiles = (int)((M + 63) / 64); p.n_nblocks = cpad / 64;
        dim3 grid((unsigned)(p.n_mtiles * p.n_nblocks), 1, splits);
        HL_CONV_GO(2, 2, 1, 1, grid);
    }
#undef HL_CONV_GO
    if (splits > 1) return finish("k_conv");
    return check_launch("k_conv");
}

static int gn_chunks(int HW, int C) {
    if ((long)HW * C <= 524288) return 1;   // one 1024-thread workgroup per image: statistics + affine in one launch
    int c = HW / 256;
    if (c < 1) c = 1;
    if (c > 128) c = 128;
    return c;
}
size_t gn_scratch_floats(int N) { return (size_t)N * 128 * 32 * 2; }

int groupnorm_coef(const View &x, const float *gamma, const float *beta, const float *emb, long emb_pitch, float *cA, float *cB,
                   float *scratch, hipStream_t st, float *gstat, float eps) {
    HL_REQUIRE(x.p && gamma && beta && cA && cB && scratch, "groupnorm_coef: null argument");
    HL_REQUIRE(x.C % 32 == 0, "GroupNorm32 needs C %% 32 == 0 (C=%d)", x.C);
    const int HW = x.H * x.W, cq = x.C / 4;
    const int nch = gn_chunks(HW, x.C);
    if (nch == 1) {
        const int cg = x.C / 32;
        dim3 grid(32, x.N);
        if (cg % 4 == 0 && ((uintptr_t)x.p % 16) == 0 && x.pitch % 4 == 0)
            hipLaunchKernelGGL(k_gn_small<4>, grid, dim3(256), 0, st, x.p, x.pitch, HW, x.C, gamma, beta, emb, emb_pitch, cA, cB, gstat, eps);
        else if (cg % 2 == 0 && ((uintptr_t)x.p % 8) == 0 && x.pitch % 2 == 0)
            hipLaunchKernelGGL(k_gn_small<2>, grid, dim3(256), 0, st, x.p, x.pitch, HW, x.C, gamma, beta, emb, emb_pitch, cA, cB, gstat, eps);
        else
            hipLaunchKernelGGL(k_gn_small<1>, grid, dim3(256), 0, st, x.p, x.pitch, HW, x.C, gamma, beta, emb, emb_pitch, cA, cB, gstat, eps);
        return check_launch("k_gn_small");
    }
    int k = (nch == 1 ? 1024 : 512) / cq;
    if (k < 1) k = 1;
    HL_REQUIRE(cq <= 1024, "groupnorm_coef: C too large");
    const int threads = cq * k > 64 ? cq * k : 64;
    const size_t shm = (size_t)2 * k * x.C * sizeof(float);
    hipLaunchKernelGGL(k_gn_partial, dim3(nch, x.N), dim3(threads), shm, st, x.p, x.pitch, HW, x.C, nch, scratch, gamma, beta, emb,
                       emb_pitch, cA, cB, gstat, eps);
    int rc = check_launch("k_gn_partial");
    if (rc || nch == 1) return rc;
    hipLaunchKernelGGL(k_gn_coef, dim3(32, x.N), dim3(64), 0, st, scratch, nch, HW, x.C, gamma, beta, emb, emb_pitch, cA, cB, gstat, eps);
    return check_launch("k_gn_coef");
}

int gn_apply(const View &x, const float *cA, const float *cB, int act, float *y, hipStream_t st) {
    const long npix = x.pixels();
    long g = (npix * (x.C / 4) + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_gn_apply, dim3((unsigned)g), dim3(256), 0, st, x.p, x.pitch, (long)x.H * x.W, npix, x.C, cA, cB, act, y);
    return check_launch("k_gn_apply");
}

int tensor_absmax(const View &x, float *amax, hipStream_t st) {
    HL_REQUIRE(x.p && amax && x.C % 4 == 0 && x.pitch % 4 == 0, "tensor_absmax: bad argument");
    const long HW = (long)x.H * x.W;
    if (hipMemsetAsync(amax, 0, (size_t)x.N * sizeof(float), st) != hipSuccess) return fail(HL_ERR_RUNTIME, "tensor_absmax: memset");
    const unsigned gx = (unsigned)std::min<long>(256, std::max<long>(1, HW * x.C / 16384));
    hipLaunchKernelGGL(k_tensor_absmax, dim3(gx, (unsigned)x.N), dim3(256), 0, st, x.p, x.pitch, HW, x.C, amax);
    return check_launch("k_tensor_absmax");
}

int tensor_totals(const View &x, float *totals, hipStream_t st) {
    HL_REQUIRE(x.p && totals && x.C % 4 == 0 && x.pitch % 4 == 0, "tensor_totals: bad argument");
    const long HW = (long)x.H * x.W;
    if (hipMemsetAsync(totals, 0, conv_stats_floats(x.N, HW) * sizeof(float), st) != hipSuccess) return fail(HL_ERR_RUNTIME, "tensor_totals: memset");
    const unsigned gx = (unsigned)std::min<long>(256, std::max<long>(1, HW * x.C / 16384));      // (>= 16 K values per workgroup: a contribution stays far below the capacity for sane inputs)
    hipLaunchKernelGGL(k_tensor_totals, dim3(gx, (unsigned)x.N), dim3(256), 0, st, x.p, x.pitch, HW, x.C, totals);
    return check_launch("k_tensor_totals");
}

int groupnorm_coef_stats(const View &x, const float *gt, const float *gamma, const float *beta, const float *emb,
                         long emb_pitch, float *cA, float *cB, hipStream_t st, float eps) {
    HL_REQUIRE(gt && gamma && beta && cA && cB, "groupnorm_coef_stats: bad argument");
    HL_REQUIRE(x.C % 32 == 0, "GroupNorm32 needs C %% 32 == 0 (C=%d)", x.C);
    static const int abl_ = [] { const char *e_ = getenv("HL_ABL_SKIP"); return e_ ? atoi(e_) : 0; }();   // TIMING ablation (wrong results)
    if (abl_ & 2) return HL_OK;
    hipLaunchKernelGGL(k_gn_coef_tot, dim3(32, x.N), dim3(64), 0, st, gt, x.H * x.W, x.C, gamma, beta, emb, emb_pitch, cA, cB, eps);
    return check_launch("k_gn_coef_tot");
}

int linear_small(const float *in, long in_pitch, int B, int K, const float *W, const float *bias, int O, int silu_in,
                 const float *addrow, const int64_t *idx, float *out, long out_pitch, hipStream_t st) {
    HL_REQUIRE(in && W && out && B >= 1, "linear_small: bad argument");
    HL_REQUIRE(B <= 16, "linear_small: batch %d > 16 (split the batch)", B);
    dim3 grid((O + 3) / 4);
    if (B <= 4)
        hipLaunchKernelGGL(k_linear_small<4>, grid, dim3(256), 0, st, in, in_pitch, B, K, W, bias, O, silu_in, addrow, idx, out, out_pitch);
    else if (B <= 8)
        hipLaunchKernelGGL(k_linear_small<8>, grid, dim3(256), 0, st, in, in_pitch, B, K, W, bias, O, silu_in, addrow, idx, out, out_pitch);
    else
        hipLaunchKernelGGL(k_linear_small<16>, grid, dim3(256), 0, st, in, in_pitch, B, K, W, bias, O, silu_in, addrow, idx, out, out_pitch);
    return check_launch("k_linear_small");
}

int timestep_embedding(const int64_t *t, const float *tf, int B, int dim, float *out, hipStream_t st) {
    HL_REQUIRE((t || tf) && out && B > 0 && dim > 0, "timestep_embedding: bad argument");
    hipLaunchKernelGGL(k_timestep_embedding, dim3((B * dim + 255) / 256), dim3(256), 0, st, t, tf, B, dim, out);
    return check_launch("k_timestep_embedding");
}

int attention(const float *qkv, int N, int T, int C, int heads, float *out, hipStream_t st, int h2, float *out_totals, int *totals_emitted) {
    HL_REQUIRE(qkv && out && heads > 0 && C % heads == 0, "attention: bad argument");
    if (totals_emitted) *totals_emitted = 0;
    static const int att_h2 = [] { const char *e_ = getenv("HL_ATT_H2"); return e_ ? atoi(e_) : 1; }();   // developer knob (read once): 0 = the fp32-MFMA kernels in every mode
    if (!att_h2) h2 = 0;
    const int ch = C / heads;
    // 32 queries per wave; pick waves per workgroup so the grid has at least ~256 workgroups (K/V tiles are
    // shared through LDS inside a workgroup, and are L2-resident across workgroups)
    const int qtiles = (T + 31) / 32;
    int wpb = 4;
    while (wpb > 1 && (long)((qtiles + wpb - 1) / wpb) * N * heads < 256) wpb >>= 1;
    dim3 grid((qtiles + wpb - 1) / wpb, N * heads);
#define HL_ATT(CH_)                                                                                         \
    do {                                                                                                    \
        if (wpb == 4) hipLaunchKernelGGL((k_attention<CH_, 4>), grid, dim3(256), 0, st, qkv, T, C, heads, out);      \
        else if (wpb == 2) hipLaunchKernelGGL((k_attention<CH_, 2>), grid, dim3(128), 0, st, qkv, T, C, heads, out); \
        else hipLaunchKernelGGL((k_attention<CH_, 1>), grid, dim3(64), 0, st, qkv, T, C, heads, out);                \
    } while (0)
    // short sequences: split the keys over the 4 waves of a workgroup instead (one query tile per workgroup)
    // (with fp16x2 products the key-split kernel also takes the 1 024-token level of batches 8 ... 16: B = 8 49.4 -> 49.0 ms per forward)
    static const long ks_max = [] { const char *e_ = getenv("HL_ATT_KS_MAX"); return e_ ? atol(e_) : 4097L; }();   // developer knob (read once)
    if ((long)qtiles * N * heads < (h2 ? std::max(ks_max, 1024L) : 1024L) && (ch == 96 || ch == 192) && (3L * C) % 4 == 0) {
        dim3 gks(qtiles, N * heads);
        if (ch == 96 && h2)
            hipLaunchKernelGGL((k_attention_ks<96, 4, true, true>), gks, dim3(256), (size_t)4 * (96 * 33 + 64) * sizeof(float), st, qkv, T, C, heads, out, out_totals, N);
        else if (ch == 96)
            hipLaunchKernelGGL((k_attention_ks<96, 4, true, false>), gks, dim3(256), (size_t)4 * (96 * 33 + 64) * sizeof(float), st, qkv, T, C,
                               heads, out, out_totals, N);
        else if (h2)
            hipLaunchKernelGGL((k_attention_ks<192, 4, false, true>), gks, dim3(256), (size_t)4 * (192 * 33 + 64) * sizeof(float), st, qkv, T, C, heads, out, out_totals, N);
        else
            hipLaunchKernelGGL((k_attention_ks<192, 4, false, false>), gks, dim3(256), (size_t)4 * (192 * 33 + 64) * sizeof(float), st, qkv, T,
                               C, heads, out, out_totals, N);
        if (totals_emitted && out_totals) *totals_emitted = 1;
        return check_launch("k_attention_ks");
    }
    switch (ch) {
        case 32: HL_ATT(32); break;
        case 64: HL_ATT(64); break;
        case 96: HL_ATT(96); break;
        case 128: HL_ATT(128); break;
        case 192: HL_ATT(192); break;
        default: {
            HL_REQUIRE((size_t)T * 4 * sizeof(float) <= 60000, "attention: T=%d too long for the generic kernel", T);
            hipLaunchKernelGGL(k_attention_generic, dim3((T + 3) / 4, N * heads), dim3(256), (size_t)T * 4 * sizeof(float), st,
                               qkv, T, C, heads, ch, out);
        }
    }
#undef HL_ATT
    return check_launch("k_attention");
}

int prep_inputs(const float *x, const float *xc, int B, int C, int H, int W, int Cpad, float *xo, float *xs, hipStream_t st, float *xo_tot, float *xs_tot) {
    HL_REQUIRE(x && xo && Cpad >= C && B > 0, "prep_inputs: bad argument");
    const long per_img = ((long)H * W * Cpad + 255) / 256;
    const unsigned gx = (unsigned)std::max<long>(1, std::min<long>(per_img, (2048 + B - 1) / B));
    hipLaunchKernelGGL(k_prep_inputs, dim3(gx, (unsigned)B), dim3(256), 0, st, x, xc, B, C, H * W, Cpad, xo, xs, xo_tot, xs_tot);
    return check_launch("k_prep_inputs");
}

int layernorm(const View &x, const float *gamma, const float *beta, float *y, hipStream_t st) {
    HL_REQUIRE(x.p && gamma && beta && y, "layernorm: null argument");
    const long npix = x.pixels();
    hipLaunchKernelGGL(k_layernorm, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, st, x.p, x.pitch, npix, x.C, gamma, beta, y);
    return check_launch("k_layernorm");
}

int geglu(const float *in, long npix, int F, float *out, hipStream_t st) {
    HL_REQUIRE(in && out, "geglu: null argument");
    long g = (npix * F + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_geglu, dim3((unsigned)g), dim3(256), 0, st, in, npix, F, out);
    return check_launch("k_geglu");
}

int add_rowvec(const View &x, const float *v, hipStream_t st, long vpitch) {
    HL_REQUIRE(x.p && v, "add_rowvec: null argument");
    long g = ((long)x.pixels() * x.C + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_add_rowvec, dim3((unsigned)g), dim3(256), 0, st, x.p, x.pitch, (long)x.H * x.W, (long)x.pixels(), x.C, v, vpitch ? vpitch : (long)x.C);
    return check_launch("k_add_rowvec");
}

int prep_inputs_3d(const float *x, const float *xc, int B, int C, int H, int W, int Cpad, float *xo, float *xs, hipStream_t st) {
    HL_REQUIRE(x && xo && Cpad >= C, "prep_inputs_3d: bad argument");
    hipLaunchKernelGGL(k_prep_inputs_3d, dim3(2048), dim3(256), 0, st, x, xc, B, C, H, W, Cpad, xo, xs);
    return check_launch("k_prep_inputs_3d");
}

int unroll_planes(const float *in, int B, int C, int H, int W, float *out, hipStream_t st) {
    HL_REQUIRE(in && out, "unroll_planes: null argument");
    hipLaunchKernelGGL(k_unroll_planes, dim3(2048), dim3(256), 0, st, in, B, C, H, W, out);
    return check_launch("k_unroll_planes");
}

int gn_apply_3d(const View &h, const float *coefA, const float *coefB, float *sums, float *y, hipStream_t st) {
    HL_REQUIRE(h.p && coefA && coefB && sums && y && h.W % 3 == 0, "gn_apply_3d: bad argument");
    const int W = h.W / 3;
    hipLaunchKernelGGL(k_plane_sums, dim3((unsigned)(h.N * 3 * (h.H + W))), dim3(256), 0, st, h.p, h.pitch, h.H, W, h.C, sums);
    int rc = check_launch("k_plane_sums");
    if (rc) return rc;
    long g = ((long)h.pixels() * h.C + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_gn_apply_3d, dim3((unsigned)g), dim3(256), 0, st, h.p, h.pitch, h.N, h.H, W, h.C, coefA, coefB, sums, y);
    return check_launch("k_gn_apply_3d");
}

}  // namespace hl
