// Backward kernels of the tri-plane UNet (UNet half), MI355X (gfx950), fp32.
//
// What the reference trains with: GaussianDiffusion.training_losses (gaussian_diffusion.py:688-772) -> MSE on the network output ->
// loss.backward() through UNetModel.forward (unet.py:550-615), driven by TrainLoop.forward_backward (train_util.py:200-285).  Autograd
// of the stock ops is replaced, op by op (humanliff_amd/improved_diffusion/unet_train.py holds the autograd.Functions), by
//   conv forward / backward-data   the FORWARD conv kernels of hl_unet_kernels.hip: backward-data of a 3x3 / 1x1 convolution is the
//                                  same convolution of the output gradient with the flipped, channel-transposed weights
//   k_conv_wgrad_t / _1x1          backward-weights + bias:  dW[co][ci][ky][kx] = sum_p dY[p][co] * X[p*stride + (ky,kx) - pad][ci]
//                                  as a GEMM over the pixels on v_mfma_f32_32x32x2_f32 (3x3 layers: _t, all taps per workgroup)
//   k_gn_apply (hl_unet_kernels)   GroupNorm32 (+scale/shift) (+SiLU) apply, nn.py:100, unet.py:198-219
//   k_gn_bwd_reduce / _apply       its backward: per-(n,c) reductions, then dx = k1*du + k2*x + k3
// No float atomics on this path: the GroupNorm reductions and the convolutions' weight / bias gradients are partial sums added in a fixed
// order (k_gn_bwd_fin, k_wgrad_finish), so the kernels of a training step give the same bits run to run.  (The atomics-based fallback
// k_conv_wgrad for channel counts that are not multiples of 4 - no layer of the network - was removed in round 4.)
#include "hl_unet_kernels.h"

namespace hl {
namespace {

__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v)); }


// ---------------------------------------------------------------------------------------------
// backward-weights of the 3x3 layers, all nine taps per workgroup (k_conv_wgrad_t + k_wgrad_finish)
// ---------------------------------------------------------------------------------------------
// The first version (k_conv_wgrad, removed) took its operands straight from L2 - 1 KiB per four MFMAs and wave - and gave every tap a workgroup of its
// own, so a 3x3 layer pulls X and dY through the L2 nine times per channel-block pair: the kernel runs at the L2's delivery rate
// (0.38 of the fp32 matrix peak over the production network).  Here a workgroup owns a (64 co x 64 ci) block of dW for ALL nine taps
// and walks a slab of 8x8-pixel output tiles (4x8 for stride 2):
//   * per tile the 64 dY rows and the (8+2)x(8+2) input patch (9x17 for stride 2) of the 64 input channels are staged in LDS ONCE
//     (42 KB; fetched one tile ahead into registers, written after the barrier that ends the previous tile's reads) - 1.56 patch pixels
//     per output pixel instead of 9;
//   * wave (i, j) owns channel quadrant (co half i, ci half j) x 9 taps = nine 32x32 accumulator tiles (144 registers, two workgroups
//     per CU).  A k-step is one pixel pair: ONE ds_read_b32 of dY (A operand: lane (m, kh) = channel m of pixel 2s + kh) and NINE of
//     the patch (B operand of tap (ky, kx): the same lane, patch pixel shifted by the tap - a compile-time LDS offset) feed nine MFMAs.
//     LDS is split by channel half, [half][pixel][32], so each 32-lane group of a read covers 32 consecutive floats (conflict-free;
//     ds_read_b32 banks over 32 lanes).  10 reads x 2 LDS cycles per 9 x 64 matrix cycles: the LDS array is ~14 % busy.
//   * zero padding / ragged tiles / the nearest-x2 upsample / stride 2 are per-lane source offsets of the staging loads, out-of-range
//     ones point outside the buffer (the hardware returns zeros).
// No atomics: every workgroup writes its block to a partial buffer [slab][co][tap][ci] (caller's scratch) and k_wgrad_finish sums the
// slabs in a fixed order into the OIHW gradient - the weight gradients of these layers are bit-reproducible run to run.  The bias
// gradient (row sums of dY, taken by the j = 0 waves of the ci-block-0 workgroups from the A operands they read anyway) goes the
// same way, as a tenth "tap".
struct WgradT {
    const float *x; long x_pitch; int N, Hin, Win, Cx;        // conv input (NHWC, Cx % 4 == 0 channels present)
    const float *dy; long dy_pitch; int Hout, Wout, Cy;       // output gradient (NHWC, Cy % 4 == 0)
    int ups;
    float *part;                                               // [slabs][n_co*64][10][n_ci*64]: taps 0..8, 9 = bias (column 0 of ci block 0)
    int n_co, n_ci, slabs;
    int tilesX, tilesY; long tiles, per_slab;                  // output tiles per image row / column, in all, per slab
    int want_b;
};

template <int S>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_t(const WgradT p) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int TH = 8 / S, TW = 8, NPIX = TH * TW;                          // output pixels of a tile
    constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, NP = PH * PW;  // input patch: 10x10 (stride 1), 9x17 (stride 2)
    constexpr int NLB = (NP + 15) / 16, NLA = NPIX / 16;                       // staging loads per thread (16 threads x float4 = 64 channels)
    __shared__ float ldsB[2][NP][32];
    __shared__ float ldsA[2][NPIX][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, m = lane & 31;
    const int wi = wave >> 1, wj = wave & 1;
    // XCD-aware order: the channel blocks of one slab (same pixels) run on the same XCD, so its L2 serves their common X / dY reads
    const int blocks = p.n_co * p.n_ci;
    const int per_xcd = (int)(gridDim.x >> 3);
    const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= blocks * p.slabs) return;
    const int slab = logical / blocks, blk = logical - slab * blocks;
    const int cob = blk / p.n_ci, cib = blk - cob * p.n_ci;
    const int co0 = cob * 64, ci0 = cib * 64;
    const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;
    const __amdgpu_buffer_rsrc_t rsY =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.dy, (short)0, (int)((long)p.N * p.Hout * p.Wout * p.dy_pitch * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.x, (short)0, (int)((long)p.N * p.Hin * p.Win * p.x_pitch * 4), 0x00020000);
    const int c4 = tid & 15, e0 = tid >> 4;
    const bool a_ok = co0 + 4 * c4 < p.Cy, b_ok = ci0 + 4 * c4 < p.Cx;
    const long t0 = (long)slab * p.per_slab, t1 = min(p.tiles, t0 + p.per_slab);
    const int tpi = p.tilesX * p.tilesY;

    f32x4 rb[NLB], ra[NLA];
    auto fetch = [&](long tile) {
        const int n = (int)(tile / tpi), r = (int)(tile - (long)n * tpi);
        const int tyb = r / p.tilesX, txb = r - tyb * p.tilesX;
        const int oy0 = tyb * TH, ox0 = txb * TW;
#pragma unroll
        for (int u = 0; u < NLB; ++u) {
            const int pos = e0 + 16 * u, pr = pos / PW, pc = pos - pr * PW;
            const int yi = oy0 * S + pr - 1, xi = ox0 * S + pc - 1;
            const bool v = pos < NP && b_ok && yi >= 0 && yi < Hv && xi >= 0 && xi < Wv;
            const int ys = p.ups ? yi >> 1 : yi, xs = p.ups ? xi >> 1 : xi;
            const unsigned off = v ? (unsigned)((((long)n * p.Hin + ys) * p.Win + xs) * p.x_pitch + ci0 + 4 * c4) * 4u : OOB;
            rb[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < NLA; ++u) {
            const int q = e0 + 16 * u, yo = oy0 + (q >> 3), xo = ox0 + (q & 7);
            const bool v = a_ok && yo < p.Hout && xo < p.Wout;
            const unsigned off = v ? (unsigned)((((long)n * p.Hout + yo) * p.Wout + xo) * p.dy_pitch + co0 + 4 * c4) * 4u : OOB;
            ra[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0));
        }
    };
    auto stage = [&]() {
        const int half = c4 >> 3, cc = (c4 & 7) * 4;
#pragma unroll
        for (int u = 0; u < NLB; ++u) {
            const int pos = e0 + 16 * u;
            if (pos < NP) *reinterpret_cast<f32x4 *>(&ldsB[half][pos][cc]) = rb[u];
        }
#pragma unroll
        for (int u = 0; u < NLA; ++u) *reinterpret_cast<f32x4 *>(&ldsA[half][e0 + 16 * u][cc]) = ra[u];
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bs = 0.f;
    const float *la = &ldsA[wi][kh][m];                          // + pixel pair s: 2 s * 32 floats
    const float *lb = &ldsB[wj][kh * S][m];                      // + patch pixel of (pair s, tap): compile-time offsets

    if (t0 < t1) fetch(t0);
    for (long tile = t0; tile < t1; ++tile) {
        if (tile > t0) __syncthreads();                          // the previous tile's reads are done
        stage();
        __syncthreads();
        if (tile + 1 < t1) fetch(tile + 1);                      // in flight during this tile's MFMAs
#pragma unroll 1
        for (int row = 0; row < TH; ++row) {                     // a tile row = 4 pixel pairs; only the row offset is a run-time value
            const float *lar = la + row * (8 * 32), *lbr = lb + row * (S * PW * 32);
#pragma unroll
            for (int sc = 0; sc < 4; ++sc) {
                const float a = lar[(2 * sc) * 32];
                float b[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) b[t] = lbr[((t / 3) * PW + 2 * sc * S + t % 3) * 32];
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
                bs += a;
            }
        }
    }
    // partial block: accumulator r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5) (co), column l & 31 (ci)
    const long CoP = (long)p.n_co * 64, CiP = (long)p.n_ci * 64;
    float *pp = p.part + (long)slab * CoP * 10 * CiP;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
            pp[((long)(co0 + 32 * wi + row) * 10 + t) * CiP + ci0 + 32 * wj + m] = acc[t][r];
        }
    if (p.want_b && cib == 0 && wj == 0) {
        bs += __shfl_xor(bs, 32);
        if (kh == 0) pp[((long)(co0 + 32 * wi + m) * 10 + 9) * CiP] = bs;
    }
}

// k_conv_wgrad_h16: k_conv_wgrad_t (stride 1, with or without the nearest-x2 upsample) with 16-bit operands and fp32 accumulation on v_mfma_f32_32x32x16_{f16,bf16} -
// the arithmetic of the 16-bit training modes (HL_CONV_FP16 / HL_CONV_BF16).  Same work split, same partial buffer and finish kernel.
// The contraction runs over PIXELS, and a 16-bit MFMA fragment wants 8 consecutive k values per lane, so both operands sit in LDS
// pixel-contiguous, [channel][pixel]: the transposition happens in registers while staging (a thread fetches a few pixels x 4 channels
// as fp32, rounds them and writes one run of pixels per channel), never in memory.
//   * dY: [co 64][8 rows][8 px], 16 bytes per tile row; a k-step of 16 pixels = two tile rows = the two lane halves of a fragment;
//   * the input patch three times, one copy per tap column kx: [kx][ci 64][10 rows][8 px] holds columns kx..kx+7 of every patch row,
//     so the fragment of tap (ky, kx) for tile row y is the aligned 16 bytes of row y + ky in copy kx;
//   * channel rows are padded by 16 bytes (144 / 176 bytes): the 16 lanes a ds_read_b128 is served in hit 16 different slots.
// Per k-step and wave: 1 + 9 ds_read_b128 for 9 MFMAs.  The kernel is bound by what it pulls through L2 (41.6 KB of fp32 per 64 pixels
// and workgroup), not by the matrix pipe - two workgroups per CU keep loads in flight behind the other's MFMAs.
template <bool F16>
__device__ __forceinline__ unsigned wg_pack2(float a, float b) {
    if constexpr (F16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 h = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, h);
    } else {
        const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
        const unsigned ra = ua + 0x7fffu + ((ua >> 16) & 1u), rb = ub + 0x7fffu + ((ub >> 16) & 1u);
        return __builtin_amdgcn_perm(rb, ra, 0x07060302);
    }
}
template <bool F16>
__device__ __forceinline__ f32x16 wg_mma(const u32x4 a, const u32x4 b, const f32x16 c) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool F16>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_h16(const WgradT p) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int AP = 144, BP = 176, A_BYTES = 64 * AP, B_COPY = 64 * BP;      // channel-row pitches (bytes), operand images
    __shared__ __attribute__((aligned(16))) char lds[A_BYTES + 3 * B_COPY];     // 42 KB: two workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, m = lane & 31;
    const int wi = wave >> 1, wj = wave & 1;
    const int blocks = p.n_co * p.n_ci;
    const int per_xcd = (int)(gridDim.x >> 3);
    const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= blocks * p.slabs) return;
    const int slab = logical / blocks, blk = logical - slab * blocks;
    const int cob = blk / p.n_ci, cib = blk - cob * p.n_ci;
    const int co0 = cob * 64, ci0 = cib * 64;
    const __amdgpu_buffer_rsrc_t rsY =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.dy, (short)0, (int)((long)p.N * p.Hout * p.Wout * p.dy_pitch * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.x, (short)0, (int)((long)p.N * p.Hin * p.Win * p.x_pitch * 4), 0x00020000);
    const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;
    const int c4 = tid & 15;
    const int ar = (tid >> 4) & 7, ah = tid >> 7;              // dY: tile row, half row (4 pixels) of this thread
    const int br = tid >> 4;                                   // patch: row of this thread (threads 0..159)
    const bool a_ok = co0 + 4 * c4 < p.Cy, b_ok = ci0 + 4 * c4 < p.Cx && br < 10;
    const long t0 = (long)slab * p.per_slab, t1 = min(p.tiles, t0 + p.per_slab);
    const int tpi = p.tilesX * p.tilesY;

    f32x4 ra[4], rb[10];
    auto fetch = [&](long tile) {
        const int n = (int)(tile / tpi), r = (int)(tile - (long)n * tpi);
        const int tyb = r / p.tilesX, txb = r - tyb * p.tilesX;
        const int oy0 = tyb * 8, ox0 = txb * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yo = oy0 + ar, xo = ox0 + 4 * ah + j;
            const bool v = a_ok && yo < p.Hout && xo < p.Wout;
            const unsigned off = v ? (unsigned)((((long)n * p.Hout + yo) * p.Wout + xo) * p.dy_pitch + co0 + 4 * c4) * 4u : OOB;
            ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0));
        }
        if (tid < 160) {
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                const int yi = oy0 + br - 1, xi = ox0 + c - 1;          // (p.ups: coordinates in the nearest-x2 image)
                const bool v = b_ok && yi >= 0 && yi < Hv && xi >= 0 && xi < Wv;
                const int ys = p.ups ? yi >> 1 : yi, xs = p.ups ? xi >> 1 : xi;
                const unsigned off = v ? (unsigned)((((long)n * p.Hin + ys) * p.Win + xs) * p.x_pitch + ci0 + 4 * c4) * 4u : OOB;
                rb[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0));
            }
        }
    };
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};                          // bias gradient: fp32 sums of this thread's dY values
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                           // channel 4 c4 + i: four pixels of tile row ar -> 8 bytes
            const unsigned lo = wg_pack2<F16>(ra[0][i], ra[1][i]), hi = wg_pack2<F16>(ra[2][i], ra[3][i]);
            *reinterpret_cast<uint2 *>(lds + (4 * c4 + i) * AP + ar * 16 + ah * 8) = uint2{lo, hi};
            bsum[i] += (ra[0][i] + ra[1][i]) + (ra[2][i] + ra[3][i]);
        }
        if (tid < 160) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {                       // channel 4 c4 + i: patch row br, columns 0..9 -> the three shifted 8-pixel runs
                unsigned e[5], o[4];
#pragma unroll
                for (int k = 0; k < 5; ++k) e[k] = wg_pack2<F16>(rb[2 * k][i], rb[2 * k + 1][i]);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_perm(e[k + 1], e[k], 0x05040302);      // (h[2k+1], h[2k+2])
                char *q = lds + A_BYTES + (4 * c4 + i) * BP + br * 16;
                *reinterpret_cast<u32x4 *>(q) = u32x4{e[0], e[1], e[2], e[3]};
                *reinterpret_cast<u32x4 *>(q + B_COPY) = u32x4{o[0], o[1], o[2], o[3]};
                *reinterpret_cast<u32x4 *>(q + 2 * B_COPY) = u32x4{e[1], e[2], e[3], e[4]};
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const char *la = lds + (32 * wi + m) * AP + g * 16;                      // + k-step s: 32 bytes (two tile rows)
    const char *lb = lds + A_BYTES + (32 * wj + m) * BP + g * 16;            // + copy kx, + (2 s + ky) * 16

    if (t0 < t1) fetch(t0);
    for (long tile = t0; tile < t1; ++tile) {
        if (tile > t0) __syncthreads();                          // the previous tile's reads are done
        stage();
        __syncthreads();
        if (tile + 1 < t1) fetch(tile + 1);                      // in flight during this tile's MFMAs
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const u32x4 a = *reinterpret_cast<const u32x4 *>(la + s * 32);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                u32x4 b[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) b[kx] = *reinterpret_cast<const u32x4 *>(lb + kx * B_COPY + (2 * s + ky) * 16);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = wg_mma<F16>(a, b[kx], acc[ky * 3 + kx]);
            }
        }
    }
    // partial block: accumulator r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5) (co), column l & 31 (ci)
    const long CoP = (long)p.n_co * 64, CiP = (long)p.n_ci * 64;
    float *pp = p.part + (long)slab * CoP * 10 * CiP;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
            pp[((long)(co0 + 32 * wi + row) * 10 + t) * CiP + ci0 + 32 * wj + m] = acc[t][r];
        }
    if (p.want_b && cib == 0) {                                  // bias slot: the 16 threads that share a channel quad meet in LDS, fixed order
        __syncthreads();
        float *red = reinterpret_cast<float *>(lds);
        *reinterpret_cast<f32x4 *>(red + ((tid >> 4) * 16 + c4) * 4) = bsum;
        __syncthreads();
        if (tid < 64) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) v += red[(k * 16 + (tid >> 2)) * 4 + (tid & 3)];
            pp[((long)(co0 + tid) * 10 + 9) * CiP] = v;
        }
    }
}

// 1x1 layers (skip / zero / qkv / proj convolutions): no halo, so a tile is 64 consecutive pixels of the flattened image batch.  With
// a single tap the only reuse there is sits in the channel block, so a workgroup owns 192 output x 64 input channels (every width of
// the network is a multiple of 192 and of 64): wave (i, j) holds co tiles 3i..3i+2 x ci tile j - three A reads and one B read per
// three MFMAs - and a tile moves 64 KB from L2 for 12 x 32 MFMAs, 2/3 more per byte than k_conv_wgrad's 64 x 64 block per wave.
// Same partial / finish scheme as k_conv_wgrad_t (one "tap" + the bias slot): deterministic.
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_1x1(const WgradT p) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NPIX = 64;
    __shared__ float ldsA[6][NPIX][32];
    __shared__ float ldsB[2][NPIX][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, m = lane & 31;
    const int wi = wave >> 1, wj = wave & 1;
    const int blocks = p.n_co * p.n_ci;
    const int per_xcd = (int)(gridDim.x >> 3);
    const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= blocks * p.slabs) return;
    const int slab = logical / blocks, blk = logical - slab * blocks;
    const int cob = blk / p.n_ci, cib = blk - cob * p.n_ci;
    const int co0 = cob * 192, ci0 = cib * 64;
    const long P = (long)p.N * p.Hout * p.Wout;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void *)p.dy, (short)0, (int)(P * p.dy_pitch * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, (short)0, (int)(P * p.x_pitch * 4), 0x00020000);
    const int c4 = tid & 15, e0 = tid >> 4;
    const bool b_ok = ci0 + 4 * c4 < p.Cx;
    const long t0 = (long)slab * p.per_slab, t1 = min(p.tiles, t0 + p.per_slab);

    f32x4 ra[12], rb[4];
    auto fetch = [&](long tile) {
        const long px0 = tile * NPIX;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long px = px0 + e0 + 16 * u;
            const bool in = px < P;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const bool v = in && co0 + 64 * g + 4 * c4 < p.Cy;
                ra[g * 4 + u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rsY, v ? (unsigned)(px * p.dy_pitch + co0 + 64 * g + 4 * c4) * 4u : OOB, 0, 0));
            }
            rb[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rsX, (in && b_ok) ? (unsigned)(px * p.x_pitch + ci0 + 4 * c4) * 4u : OOB, 0, 0));
        }
    };
    auto stage = [&]() {
        const int half = c4 >> 3, cc = (c4 & 7) * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int g = 0; g < 3; ++g) *reinterpret_cast<f32x4 *>(&ldsA[2 * g + half][e0 + 16 * u][cc]) = ra[g * 4 + u];
            *reinterpret_cast<f32x4 *>(&ldsB[half][e0 + 16 * u][cc]) = rb[u];
        }
    };
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bs[3] = {0.f, 0.f, 0.f};
    const float *la = &ldsA[3 * wi][kh][m], *lb = &ldsB[wj][kh][m];

    if (t0 < t1) fetch(t0);
    for (long tile = t0; tile < t1; ++tile) {
        if (tile > t0) __syncthreads();
        stage();
        __syncthreads();
        if (tile + 1 < t1) fetch(tile + 1);
#pragma unroll 8
        for (int s = 0; s < NPIX / 2; ++s) {
            const float b = lb[(2 * s) * 32];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float a = la[(t * NPIX + 2 * s) * 32];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                bs[t] += a;
            }
        }
    }
    const long CoP = (long)p.n_co * 192, CiP = (long)p.n_ci * 64;
    float *pp = p.part + (long)slab * CoP * 2 * CiP;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
            pp[((long)(co0 + 32 * (3 * wi + t) + row) * 2) * CiP + ci0 + 32 * wj + m] = acc[t][r];
        }
    if (p.want_b && cib == 0 && wj == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float v = bs[t] + __shfl_xor(bs[t], 32);
            if (kh == 0) pp[((long)(co0 + 32 * (3 * wi + t) + m) * 2 + 1) * CiP] = v;
        }
    }
}

// dW[co][ci][tap] = sum over the slabs in a fixed order; db[co] likewise from the extra "tap" (column 0 of ci block 0).
// A block = 64 (co, ci) pairs x 4 slab groups (group g sums slabs g, g+4, ...; the four meet in LDS in group order): four times the
// loads in flight of a thread-per-pair walk over all slabs, still a fixed order.
template <int TAPS>
__global__ __launch_bounds__(256) void k_wgrad_finish(const float *__restrict__ part, int slabs, long CoP, long CiP, int Cout, int Cin,
                                                      float *__restrict__ dw, float *__restrict__ db) {
    __shared__ float red[3][TAPS + 1][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + lane;
    const bool ok = i < (long)Cout * Cin;
    const int co = ok ? (int)(i / Cin) : 0, ci = ok ? (int)(i - (long)co * Cin) : 0;
    float v[TAPS + 1];
#pragma unroll
    for (int t = 0; t <= TAPS; ++t) v[t] = 0.f;
    const long sstride = CoP * (TAPS + 1) * CiP;
    const bool wb = db && ci == 0;
    if (ok) {
        const float *q = part + (long)co * (TAPS + 1) * CiP + ci + grp * sstride;
        for (int s = grp; s < slabs; s += 4, q += 4 * sstride) {
#pragma unroll
            for (int t = 0; t < TAPS; ++t) v[t] += q[t * CiP];
            if (wb) v[TAPS] += q[TAPS * CiP];
        }
    }
    if (grp > 0) {
#pragma unroll
        for (int t = 0; t <= TAPS; ++t) red[grp - 1][t][lane] = v[t];
    }
    __syncthreads();
    if (grp == 0 && ok) {
#pragma unroll
        for (int t = 0; t <= TAPS; ++t) v[t] = ((v[t] + red[0][t][lane]) + red[1][t][lane]) + red[2][t][lane];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) dw[i * TAPS + t] = v[t];
        if (wb) db[co] = v[TAPS];
    }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm (+SiLU) backward
// ---------------------------------------------------------------------------------------------
// forward: u = A[n,c] * x + B[n,c]  (A, B fold GroupNorm32 statistics, affine and the ResBlock's scale / shift), out = act ? silu(u) : u.
// With du = dout * (act ? silu'(u) : 1) the two reductions everything else follows from are, per (n, c):
//     S1 = sum_p du,    S2 = sum_p du * x
// (parameter gradients, the scale / shift gradients and the three per-(n,c) coefficients of dx = k1*du + k2*x + k3 are small (N,C)
// tensor algebra done by the caller).  grid (chunks, N); a thread owns a float4 of channels.  No atomics: the k pixel rows of a workgroup
// meet in LDS and are added in row order, every workgroup stores its 2C sums to part[n][chunk][2C], and k_gn_bwd_fin adds the chunks
// in chunk order - the same bits on every run.
__global__ void k_gn_bwd_reduce(const float *__restrict__ x, long x_pitch, const float *__restrict__ dout, long d_pitch, int HW, int C,
                                int nchunks, const float *__restrict__ cA, const float *__restrict__ cB, int act, float *__restrict__ part) {
    const int cq = C >> 2;
    const int k = blockDim.x / cq;
    const int c4 = threadIdx.x % cq, prow = threadIdx.x / cq;
    const int n = blockIdx.y, chunk = blockIdx.x;
    extern __shared__ float sh[];                                 // [k][2C]: row r, channel c -> (S1, S2) at r*2C + 2c
    const int per = (HW + nchunks - 1) / nchunks;
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    const f32x4 a = *reinterpret_cast<const f32x4 *>(cA + (long)n * C + c4 * 4), b = *reinterpret_cast<const f32x4 *>(cB + (long)n * C + c4 * 4);
    f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto accum = [&](const f32x4 xv, f32x4 dv) {
        if (act) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float u = a[i] * xv[i] + b[i], sg = sigmoid_f(u);
                dv[i] *= sg * (1.f + u * (1.f - sg));
            }
        }
        s1 += dv;
        s2 += dv * xv;
    };
    auto ldx = [&](int pp) { return *reinterpret_cast<const f32x4 *>(x + ((long)n * HW + pp) * x_pitch + c4 * 4); };
    auto ldd = [&](int pp) { return *reinterpret_cast<const f32x4 *>(dout + ((long)n * HW + pp) * d_pitch + c4 * 4); };
    int pp = p0 + prow;
    for (; pp + 3 * k < p1; pp += 4 * k) {                         // four pixels (eight 16-byte loads) in flight per thread
        const f32x4 x0 = ldx(pp), d0 = ldd(pp), x1 = ldx(pp + k), d1 = ldd(pp + k), x2 = ldx(pp + 2 * k), d2 = ldd(pp + 2 * k),
                    x3 = ldx(pp + 3 * k), d3 = ldd(pp + 3 * k);
        accum(x0, d0); accum(x1, d1); accum(x2, d2); accum(x3, d3);
    }
    for (; pp < p1; pp += k) accum(ldx(pp), ldd(pp));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sh[prow * 2 * C + (c4 * 4 + i) * 2] = s1[i];
        sh[prow * 2 * C + (c4 * 4 + i) * 2 + 1] = s2[i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += blockDim.x) {
        float v = sh[e];
        for (int r = 1; r < k; ++r) v += sh[r * 2 * C + e];
        part[((long)n * nchunks + chunk) * 2 * C + e] = v;
    }
}

// S[n][e] = sum over chunks of part[n][chunk][e], e < 2C: workgroup = 64 entries x 4 chunk quarters, the quarters added in order
__global__ __launch_bounds__(256) void k_gn_bwd_fin(const float *__restrict__ part, int nchunks, int C2, float *__restrict__ S) {
    __shared__ float sh[4][64];
    const int n = blockIdx.y, e = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int per = (nchunks + 3) / 4, c0 = q * per, c1 = min(nchunks, c0 + per);
    float v = 0.f;
    if (e < C2)
        for (int c = c0; c < c1; ++c) v += part[((long)n * nchunks + c) * C2 + e];
    sh[q][threadIdx.x & 63] = v;
    __syncthreads();
    if (q == 0 && e < C2) S[(long)n * C2 + e] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// dx = k1[n,c] * du + k2[n,c] * x + k3[n,c]  (+ dx_add: the gradient arriving over the residual branch), du as above
__global__ void k_gn_bwd_apply(const float *__restrict__ x, long x_pitch, const float *__restrict__ dout, long d_pitch, long pixels_per_img,
                               long npix, int C, const float *__restrict__ cA, const float *__restrict__ cB, int act,
                               const float *__restrict__ k1, const float *__restrict__ k2, const float *__restrict__ k3,
                               const float *__restrict__ dx_add, long add_pitch, float *__restrict__ dx, long dx_pitch) {
    const int cq = C >> 2;
    const long n4 = npix * cq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cq;
        const int c = (int)(i - pix * cq) * 4;
        const long n = pix / pixels_per_img;
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + pix * x_pitch + c);
        f32x4 dv = *reinterpret_cast<const f32x4 *>(dout + pix * d_pitch + c);
        if (act) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(cA + n * C + c), b = *reinterpret_cast<const f32x4 *>(cB + n * C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = a[j] * xv[j] + b[j], sg = sigmoid_f(u);
                dv[j] *= sg * (1.f + u * (1.f - sg));
            }
        }
        const f32x4 q1 = *reinterpret_cast<const f32x4 *>(k1 + n * C + c), q2 = *reinterpret_cast<const f32x4 *>(k2 + n * C + c),
                    q3 = *reinterpret_cast<const f32x4 *>(k3 + n * C + c);
        f32x4 o = q1 * dv + q2 * xv + q3;
        if (dx_add) o += *reinterpret_cast<const f32x4 *>(dx_add + pix * add_pitch + c);
        *reinterpret_cast<f32x4 *>(dx + pix * dx_pitch + c) = o;
    }
}

// The (N,C)-sized algebra between the two passes, one wave per group: with x_hat = (x - mean) * rstd, g = gamma * (1 + scale),
// D = g * du:   dx = rstd * (D - mean_grp(D) - x_hat * mean_grp(D * x_hat))  =  k1 * du + k2 * x + k3   with
//     k1 = rstd * g,   k2 = -rstd^2 * M2,   k3 = -rstd * M1 + rstd^2 * mean * M2,   M1 = sum_c g S1 / m,  M2 = sum_c g xhS / m,
//     xhS = sum_p du * x_hat = rstd * (S2 - mean * S1),   m = (C/32) * HW;
// d gamma = sum_n (1 + scale) xhS,  d beta = sum_n (1 + scale) S1,  d scale = gamma * xhS + beta * S1,  d shift = S1.
__global__ __launch_bounds__(64) void k_gn_bwd_coef(const float *__restrict__ S, const float *__restrict__ gstat, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, const float *__restrict__ ss, int N, int HW, int C,
                                                    float *__restrict__ k1, float *__restrict__ k2, float *__restrict__ k3,
                                                    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dss) {
    const int g = blockIdx.x, lane = threadIdx.x, cg = C / 32;
    const float m = (float)cg * (float)HW;
    for (int c0 = 0; c0 < cg; c0 += 64) {                          // (cg <= 64 for every C <= 2048: one trip)
        const int j = c0 + lane, c = g * cg + j;
        const bool on = j < cg;
        float dg = 0.f, dbt = 0.f;
        for (int n = 0; n < N; ++n) {
            const float mean = gstat[((long)n * 32 + g) * 2], rstd = gstat[((long)n * 32 + g) * 2 + 1];
            // group sums over ALL channels of the group (also when cg > 64: every lane walks its strided share)
            float a1 = 0.f, a2 = 0.f;
            for (int jj = lane; jj < cg; jj += 64) {
                const int cc = g * cg + jj;
                const float s1 = S[((long)n * C + cc) * 2], s2 = S[((long)n * C + cc) * 2 + 1];
                const float gh = gamma[cc] * (ss ? 1.f + ss[(long)n * 2 * C + cc] : 1.f);
                a1 += gh * s1;
                a2 += gh * (rstd * (s2 - mean * s1));
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { a1 += __shfl_xor(a1, d); a2 += __shfl_xor(a2, d); }
            const float M1 = a1 / m, M2 = a2 / m;
            if (on) {
                const float s1 = S[((long)n * C + c) * 2], s2 = S[((long)n * C + c) * 2 + 1];
                const float one_s = ss ? 1.f + ss[(long)n * 2 * C + c] : 1.f;
                const float xhS = rstd * (s2 - mean * s1);
                k1[(long)n * C + c] = rstd * gamma[c] * one_s;
                k2[(long)n * C + c] = -rstd * rstd * M2;
                k3[(long)n * C + c] = -rstd * M1 + rstd * rstd * mean * M2;
                dg += one_s * xhS;
                dbt += one_s * s1;
                if (dss) {
                    dss[(long)n * 2 * C + c] = gamma[c] * xhS + beta[c] * s1;
                    dss[(long)n * 2 * C + C + c] = s1;
                }
            }
        }
        if (on) { dgamma[c] = dg; dbeta[c] = dbt; }
    }
}

// nearest x2 upsample backward (unet.py:77): dx[n, y, x, c] = sum of the 2x2 block of d(upsampled)
__global__ void k_upsample2_bwd(const float *__restrict__ du, int N, int H, int W, int C, float *__restrict__ dx) {
    const int cq = C >> 2;
    const long n4 = (long)N * H * W * cq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cq;
        const int c = (int)(i - pix * cq) * 4;
        const int x = (int)(pix % W);
        const long t = pix / W;
        const int y = (int)(t % H);
        const long n = t / H;
        const float *q = du + (((n * 2 * H + 2 * y) * 2 * W) + 2 * x) * (long)C + c;
        const f32x4 s = (*reinterpret_cast<const f32x4 *>(q) + *reinterpret_cast<const f32x4 *>(q + C)) +
                        (*reinterpret_cast<const f32x4 *>(q + 2L * W * C) + *reinterpret_cast<const f32x4 *>(q + 2L * W * C + C));
        *reinterpret_cast<f32x4 *>(dx + pix * C + c) = s;
    }
}

// stride-2 convolution backward-data helper: z (N, 2Ho, 2Wo, C) with z[2y][2x] = dy[y][x], zeros elsewhere (then a flipped 3x3 conv of z)
__global__ void k_zero_stuff2(const float *__restrict__ dy, int N, int Ho, int Wo, int C, float *__restrict__ z) {
    const int cq = C >> 2;
    const long n4 = (long)N * 2 * Ho * 2 * Wo * cq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cq;
        const int c = (int)(i - pix * cq) * 4;
        const int x = (int)(pix % (2 * Wo));
        const long t = pix / (2 * Wo);
        const int y = (int)(t % (2 * Ho));
        const long n = t / (2 * Ho);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(x & 1) && !(y & 1)) v = *reinterpret_cast<const f32x4 *>(dy + (((n * Ho + (y >> 1)) * Wo) + (x >> 1)) * (long)C + c);
        *reinterpret_cast<f32x4 *>(z + pix * C + c) = v;
    }
}

}  // namespace
}  // namespace hl

using namespace hl;

extern "C" {

// k_conv_wgrad_t geometry: 8x8 output tiles (4x8 for stride 2); k_conv_wgrad_1x1: 64 consecutive pixels; ~512 workgroups (two per CU,
// one round)
static void wgrad_t_plan(int N, int Hout, int Wout, int Cout, int Cin, int ks, int stride, WgradT &p) {
    if (ks == 3) {
        const int TH = 8 / stride;
        p.tilesX = (Wout + 7) / 8; p.tilesY = (Hout + TH - 1) / TH;
        p.tiles = (long)N * p.tilesX * p.tilesY;
        p.n_co = (Cout + 63) / 64; p.n_ci = (Cin + 63) / 64;
    } else {
        p.tilesX = p.tilesY = 0;
        p.tiles = ((long)N * Hout * Wout + 63) / 64;
        p.n_co = (Cout + 191) / 192; p.n_ci = (Cin + 63) / 64;
    }
    const long blocks = (long)p.n_co * p.n_ci;
    long slabs = 512 / blocks;
    if (slabs < 1) slabs = 1;
    if (slabs > p.tiles) slabs = p.tiles;
    p.per_slab = (p.tiles + slabs - 1) / slabs;
    p.slabs = (int)((p.tiles + p.per_slab - 1) / p.per_slab);
}
static size_t wgrad_t_scratch(const WgradT &p, int ks) {
    return (size_t)p.slabs * p.n_co * (ks == 3 ? 64 : 192) * (ks * ks + 1) * p.n_ci * 64 * sizeof(float);
}

static bool wgrad_t_applies(int Cx, int Cy, int ks, int stride, int upsample) {
    return Cx % 4 == 0 && Cy % 4 == 0 && (ks == 3 || (ks == 1 && stride == 1 && !upsample));
}

size_t hl_conv2d_wgrad_scratch_bytes(int N, int H, int W, int Cx, int Cy, int ks, int stride, int upsample, int Cout, int Cin) {
    if (!wgrad_t_applies(Cx, Cy, ks, stride, upsample) || (stride != 1 && stride != 2)) return 0;
    const int pad = ks / 2, Hv = upsample ? 2 * H : H, Wv = upsample ? 2 * W : W;
    WgradT p{};
    wgrad_t_plan(N, (Hv + 2 * pad - ks) / stride + 1, (Wv + 2 * pad - ks) / stride + 1, Cout, Cin, ks, stride, p);
    return wgrad_t_scratch(p, ks);
}

int hl_conv2d_wgrad_nhwc_ws(const float *x, int N, int H, int W, int Cx, const float *dy, int Cy, int ks, int stride, int upsample,
                            float *dw, int Cout, int Cin, float *db, void *scratch, size_t scratch_bytes, void *stream) {
    return hl_conv2d_wgrad_nhwc_ws_mode(HL_CONV_FP32, x, N, H, W, Cx, dy, Cy, ks, stride, upsample, dw, Cout, Cin, db, scratch, scratch_bytes, stream);
}

int hl_conv2d_wgrad_nhwc_ws_mode(int conv_mode, const float *x, int N, int H, int W, int Cx, const float *dy, int Cy, int ks, int stride, int upsample,
                                 float *dw, int Cout, int Cin, float *db, void *scratch, size_t scratch_bytes, void *stream) {
    const bool h16 = (conv_mode == HL_CONV_FP16 || conv_mode == HL_CONV_BF16) && ks == 3 && stride == 1;
    HL_REQUIRE(Cx % 4 == 0 && Cy % 4 == 0, "hl_conv2d_wgrad_nhwc_ws: channel counts (x %d, dy %d) must be multiples of 4 (the atomics-based "
               "fallback for other counts is gone: every kernel of the training step sums in a fixed order)", Cx, Cy);
    HL_REQUIRE(wgrad_t_applies(Cx, Cy, ks, stride, upsample), "hl_conv2d_wgrad_nhwc_ws: a %dx%d convolution with stride %d%s has no weight-gradient kernel "
               "(3x3: stride 1 / 2 / behind an upsample; 1x1: stride 1 only - the UNet has no other)", ks, ks, stride, upsample ? " behind an upsample" : "");
    HL_REQUIRE(x && dy && dw, "hl_conv2d_wgrad_nhwc_ws: null argument");
    HL_REQUIRE(stride == 1 || (stride == 2 && !upsample), "hl_conv2d_wgrad_nhwc_ws: stride");
    HL_REQUIRE(Cin <= Cx && Cout <= Cy, "hl_conv2d_wgrad_nhwc_ws: channel counts (x %d, dy %d) must cover the weight (%d, %d)", Cx, Cy, Cout, Cin);
    const int pad = ks / 2, Hv = upsample ? 2 * H : H, Wv = upsample ? 2 * W : W;
    WgradT p{};
    p.x = x; p.x_pitch = Cx; p.N = N; p.Hin = H; p.Win = W; p.Cx = Cx;
    p.dy = dy; p.dy_pitch = Cy; p.Hout = (Hv + 2 * pad - ks) / stride + 1; p.Wout = (Wv + 2 * pad - ks) / stride + 1; p.Cy = Cy;
    p.ups = upsample; p.want_b = db != nullptr;
    HL_REQUIRE((long)N * p.Hout * p.Wout * Cy * 4 < (1L << 31) && (long)N * H * W * Cx * 4 < (1L << 31),
               "hl_conv2d_wgrad_nhwc_ws: tensors of 2 GiB and more are not addressed");
    wgrad_t_plan(N, p.Hout, p.Wout, Cout, Cin, ks, stride, p);
    const size_t need = wgrad_t_scratch(p, ks);
    HL_REQUIRE(scratch && scratch_bytes >= need, "hl_conv2d_wgrad_nhwc_ws: scratch too small (%zu bytes, hl_conv2d_wgrad_scratch_bytes says %zu)",
               scratch_bytes, need);
    p.part = static_cast<float *>(scratch);
    const unsigned grid = (unsigned)(((long)p.n_co * p.n_ci * p.slabs + 7) / 8 * 8);
    if (ks == 1) hipLaunchKernelGGL(k_conv_wgrad_1x1, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else if (h16 && conv_mode == HL_CONV_FP16) hipLaunchKernelGGL(k_conv_wgrad_h16<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else if (h16) hipLaunchKernelGGL(k_conv_wgrad_h16<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else if (stride == 1) hipLaunchKernelGGL(k_conv_wgrad_t<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_conv_wgrad_t<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    int rc = check_launch("k_conv_wgrad_t");
    if (rc) return rc;
    const long n = (long)Cout * Cin;
    if (ks == 1)
        hipLaunchKernelGGL(k_wgrad_finish<1>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, p.part, p.slabs, (long)p.n_co * 192,
                           (long)p.n_ci * 64, Cout, Cin, dw, db);
    else
        hipLaunchKernelGGL(k_wgrad_finish<9>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, p.part, p.slabs, (long)p.n_co * 64,
                           (long)p.n_ci * 64, Cout, Cin, dw, db);
    return check_launch("k_wgrad_finish");
}

int hl_gn_apply_nhwc(const float *x, long x_pitch, int N, int HW, int C, const float *coefA, const float *coefB, int silu, float *y,
                     void *stream) {
    HL_REQUIRE(x && coefA && coefB && y && C % 4 == 0 && x_pitch % 4 == 0, "hl_gn_apply_nhwc: bad argument");
    View v; v.p = const_cast<float *>(x); v.N = N; v.H = HW; v.W = 1; v.C = C; v.pitch = x_pitch;
    return gn_apply(v, coefA, coefB, silu, y, (hipStream_t)stream);
}

static int gn_bwd_chunks(int HW, int C) {
    const int cq = C / 4;
    int k = 256 / cq; if (k < 1) k = 1;
    int chunks = HW / (k * 16); if (chunks < 1) chunks = 1; if (chunks > 256) chunks = 256;
    return chunks;
}

size_t hl_gn_backward_scratch_bytes(int N, int HW, int C) { return (size_t)N * gn_bwd_chunks(HW, C) * 2 * C * sizeof(float); }

int hl_gn_backward_reduce(const float *x, long x_pitch, const float *dout, int N, int HW, int C, const float *coefA, const float *coefB,
                          int silu, float *S, void *scratch, size_t scratch_bytes, void *stream) {
    HL_REQUIRE(x && dout && coefA && coefB && S && C % 4 == 0 && x_pitch % 4 == 0 && C <= 4096, "hl_gn_backward_reduce: bad argument");
    HL_REQUIRE(scratch && scratch_bytes >= hl_gn_backward_scratch_bytes(N, HW, C),
               "hl_gn_backward_reduce: scratch too small (%zu bytes, hl_gn_backward_scratch_bytes says %zu)", scratch_bytes,
               hl_gn_backward_scratch_bytes(N, HW, C));
    const int cq = C / 4;
    int k = 256 / cq; if (k < 1) k = 1;
    const int threads = cq * k, chunks = gn_bwd_chunks(HW, C);
    float *part = static_cast<float *>(scratch);
    hipLaunchKernelGGL(k_gn_bwd_reduce, dim3(chunks, N), dim3(threads), (size_t)k * 2 * C * sizeof(float), (hipStream_t)stream, x, x_pitch, dout,
                       (long)C, HW, C, chunks, coefA, coefB, silu, part);
    int rc = check_launch("k_gn_bwd_reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(k_gn_bwd_fin, dim3((2 * C + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, part, chunks, 2 * C, S);
    return check_launch("k_gn_bwd_fin");
}

int hl_gn_backward_apply(const float *x, long x_pitch, const float *dout, int N, int HW, int C, const float *coefA, const float *coefB,
                         int silu, const float *k1, const float *k2, const float *k3, const float *dx_add, float *dx, void *stream) {
    HL_REQUIRE(x && dout && coefA && coefB && k1 && k2 && k3 && dx && C % 4 == 0 && x_pitch % 4 == 0, "hl_gn_backward_apply: bad argument");
    const long npix = (long)N * HW;
    long g = (npix * (C / 4) + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_gn_bwd_apply, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, x_pitch, dout, (long)C, (long)HW, npix, C, coefA,
                       coefB, silu, k1, k2, k3, dx_add, (long)C, dx, (long)C);
    return check_launch("k_gn_bwd_apply");
}

int hl_groupnorm_train_forward(const float *x, int N, int H, int W, int C, const float *gamma, const float *beta, const float *scale_shift,
                               int silu, float *coefA, float *coefB, float *gstat, float *y, void *scratch, size_t scratch_bytes, void *stream) {
    HL_REQUIRE(x && gamma && beta && coefA && coefB && gstat && y, "hl_groupnorm_train_forward: null argument");
    HL_REQUIRE(scratch && scratch_bytes >= gn_scratch_floats(N) * sizeof(float), "hl_groupnorm_train_forward: scratch too small");
    View v; v.p = const_cast<float *>(x); v.N = N; v.H = H; v.W = W; v.C = C; v.pitch = C;
    int rc = groupnorm_coef(v, gamma, beta, scale_shift, 2L * C, coefA, coefB, static_cast<float *>(scratch), (hipStream_t)stream, gstat);
    if (rc) return rc;
    return gn_apply(v, coefA, coefB, silu, y, (hipStream_t)stream);
}

int hl_groupnorm_train_backward(const float *x, const float *dout, int N, int H, int W, int C, const float *coefA, const float *coefB, int silu,
                                const float *gstat, const float *gamma, const float *beta, const float *scale_shift, float *dx,
                                float *dgamma, float *dbeta, float *dscale_shift, void *scratch, size_t scratch_bytes, void *stream) {
    HL_REQUIRE(x && dout && coefA && coefB && gstat && gamma && beta && dgamma && dbeta, "hl_groupnorm_train_backward: null argument");
    HL_REQUIRE(C % 32 == 0 && (scale_shift == nullptr) == (dscale_shift == nullptr), "hl_groupnorm_train_backward: bad argument");
    const size_t part_bytes = hl_gn_backward_scratch_bytes(N, H * W, C);
    HL_REQUIRE(scratch && scratch_bytes >= (size_t)N * C * 5 * sizeof(float) + part_bytes,
               "hl_groupnorm_train_backward: scratch too small (%zu bytes; N*C*5 floats + hl_gn_backward_scratch_bytes = %zu)", scratch_bytes,
               (size_t)N * C * 5 * sizeof(float) + part_bytes);
    float *S = static_cast<float *>(scratch), *k1 = S + (size_t)N * C * 2, *k2 = k1 + (size_t)N * C, *k3 = k2 + (size_t)N * C;
    int rc = hl_gn_backward_reduce(x, C, dout, N, H * W, C, coefA, coefB, silu, S, k3 + (size_t)N * C, part_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_gn_bwd_coef, dim3(32), dim3(64), 0, (hipStream_t)stream, S, gstat, gamma, beta, scale_shift, N, H * W, C, k1, k2, k3, dgamma,
                       dbeta, dscale_shift);
    rc = check_launch("k_gn_bwd_coef");
    if (rc || !dx) return rc;
    return hl_gn_backward_apply(x, C, dout, N, H * W, C, coefA, coefB, silu, k1, k2, k3, nullptr, dx, stream);
}

int hl_upsample2_backward_nhwc(const float *d_up, int N, int H, int W, int C, float *dx, void *stream) {
    HL_REQUIRE(d_up && dx && C % 4 == 0, "hl_upsample2_backward_nhwc: bad argument");
    long g = ((long)N * H * W * (C / 4) + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_upsample2_bwd, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, d_up, N, H, W, C, dx);
    return check_launch("k_upsample2_bwd");
}

int hl_zero_stuff2_nhwc(const float *dy, int N, int Ho, int Wo, int C, float *z, void *stream) {
    HL_REQUIRE(dy && z && C % 4 == 0, "hl_zero_stuff2_nhwc: bad argument");
    long g = ((long)N * 4 * Ho * Wo * (C / 4) + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_zero_stuff2, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dy, N, Ho, Wo, C, z);
    return check_launch("k_zero_stuff2");
}

}  // extern "C"
