// GroupNorm statistics as fixed-point GROUP totals (device side; included by the .hip files only).
//
// Every kernel that stores a tensor a GroupNorm will normalise (nn.py:17-19) adds, per image and GROUP OF THE NORMALISED VIEW, the
// (sum x, sum x^2) of the values it stored to two 64-bit integers with integer atomics ([shard][N][32][2]).  Integer addition is
// associative: the totals - and with them the normalisation coefficients - are bit-identical from run to run whatever the order in which
// workgroups retire, with no slot buffers and no fold kernel; a consumer needs ONE read of its image's 32 pairs to form its coefficients
// (coef_to_lds), so no coefficient launch sits between producer and consumer.  The producer is told the grouping of the view its tensor is
// normalised in (ConvK::st_cg channels per group, st_c0 = the tensor's first channel inside the view: the two producers of a decoder
// "concat" add into the SAME 32 groups).  Scales: sum 2^30 (resolution 1e-9; capacity +-8.6e9 per group and image), sum of squares 2^24
// (resolution 6e-8; capacity 2.7e11).  A contribution that is not finite or beyond the capacity sets bit 62 of the sum-of-squares word:
// consumers turn that (and an overflowed total) into NaN coefficients.  Limits, stated plainly: the reference's fp32 GroupNorm keeps working where these
// totals give up - a tensor whose sum x^2 per (image, group, shard) exceeds 2.7e11 (rms ~1e3 at the 256-pixel level with 6 channels per group) normalises to NaN here,
// loudly.  One hole remains: a total can wrap past 2^64 unnoticed if more than 1.1e12 of sum x^2 arrives in pieces that each stay under the 1e11 cap (eleven or more
// workgroups at rms ~2e3 and beyond) and the wrapped value happens to land under 2^62; closing it needs a returning atomic per contribution (its latency at the end of
// every producer workgroup) or a third word, and was not built.
#pragma once
#include <hip/hip_runtime.h>

#include "hl_unet_kernels.h"

namespace hl {

constexpr double STAT_SC_SUM = 1073741824.0;   // 2^30
constexpr double STAT_SC_SQ = 16777216.0;      // 2^24
constexpr unsigned long long STAT_POISON = 1ull << 62;

__device__ __forceinline__ unsigned long long *stat_word(float *st, int N, long img, int g, long HW) {
    const int shard = blockIdx.x & (stat_shards(HW) - 1);
    return reinterpret_cast<unsigned long long *>(st) + (((long)shard * N + img) * 32 + g) * 2;
}
// contribution (s, q) = (sum, sum of squares) of some stored values of group g of image img (of N); HW = pixels per image.  On the levels
// with many workgroups per image the totals are kept in stat_shards(HW) copies (the workgroup picks one by its index: atomics on one word
// serialise); consumers add the copies up - exact integers.
__device__ __forceinline__ void stat_add(float *st, int N, long img, int g, long HW, float s, float q) {
#ifdef HL_STAT_ABL   // timing ablation: no atomics (wrong results)
    if (N != 12345) return;
#endif
    unsigned long long *t = stat_word(st, N, img, g, HW);
    if (!(q < 1.0e11f)) { atomicOr(t + 1, STAT_POISON); return; }
    atomicAdd(t, (unsigned long long)__double2ll_rn((double)s * STAT_SC_SUM));
    atomicAdd(t + 1, (unsigned long long)__double2ll_rn((double)q * STAT_SC_SQ));
}

// The lanes of a wave hold the (s, q) of CONSECUTIVE channels (g = the group of the lane's channel, non-decreasing over the lanes that
// take part; lanes that do not pass take = false): a segmented scan adds up every run of equal groups in a fixed order and the last lane of
// a run issues ONE atomic pair for it - atomics on one word serialise (~12 ns each), 6 to 48 channels share a group.
__device__ __forceinline__ void stat_add_run(float *st, int N, long img, int g, bool take, long HW, float s, float q) {
    const int lane = threadIdx.x & 63;
    const unsigned long long act = __ballot(1);    // (the caller may have left some lanes behind: their registers are not read)
    const int key = take ? g : -1 - lane;          // (lanes outside never merge with anybody)
    if (!take) { s = 0.f; q = 0.f; }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float sp = __shfl_up(s, d), qp = __shfl_up(q, d);
        const int kp = __shfl_up(key, d);
        if (lane >= d && ((act >> (lane - d)) & 1) && kp == key) { s += sp; q += qp; }
    }
    const int kn = __shfl_down(key, 1);
    const bool next_same = lane < 63 && ((act >> (lane + 1)) & 1) && kn == key;
    if (take && !next_same) stat_add(st, N, img, g, HW, s, q);
}

// Workgroup-level pre-reduction: sub-part `sub` (a wave / round) deposits the (s, q) of local channel lc in `red` (2 * NSUB * NC floats of
// LDS nobody else uses); after the caller's barrier wg_group_flush adds a channel's NSUB parts in a fixed order, sums the channels of a
// group with 64-bit LDS atomics (exact) and issues ONE global atomic pair per group the workgroup touches.  lg = 2 * NC 64-bit words of LDS.
template <int NC>
__device__ __forceinline__ void wg_stat_put(float *red, int sub, int lc, float s, float q) {
    *reinterpret_cast<float2 *>(red + (sub * NC + lc) * 2) = make_float2(s, q);
}
template <int NC, int NSUB>
__device__ __forceinline__ void wg_group_flush(const float *red, unsigned long long *lg, float *st, int N, long img, int Cout, int n0, int c0, int cg,
                                               long HW, int tid) {
    const int g0 = (c0 + n0) / cg;
    if (tid < 2 * NC) lg[tid] = 0;
    __syncthreads();
    if (tid < NC && n0 + tid < Cout) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const float2 v = *reinterpret_cast<const float2 *>(red + (u * NC + tid) * 2);
            s += v.x;
            q += v.y;
        }
        const int gl = (c0 + n0 + tid) / cg - g0;
        if (!(q < 1.0e11f)) atomicOr(lg + gl * 2 + 1, STAT_POISON);
        else {
            atomicAdd(lg + gl * 2, (unsigned long long)__double2ll_rn((double)s * STAT_SC_SUM));
            atomicAdd(lg + gl * 2 + 1, (unsigned long long)__double2ll_rn((double)q * STAT_SC_SQ));
        }
    }
    __syncthreads();
    const int nlast = (n0 + NC < Cout ? n0 + NC : Cout) - 1, ng = (c0 + nlast) / cg - g0 + 1;
#ifdef HL_STAT_ABL
    if (N != 12345) { __syncthreads(); return; }
#endif
    if (tid < ng) {
        unsigned long long *t = stat_word(st, N, img, g0 + tid, HW);
        const unsigned long long Sv = lg[tid * 2], Qv = lg[tid * 2 + 1];
        if (Qv >= STAT_POISON) atomicOr(t + 1, STAT_POISON);
        else { atomicAdd(t, Sv); atomicAdd(t + 1, Qv); }
    }
    __syncthreads();   // (lg may be reused by the next flush)
}

// group totals (all shards) of (image n, group g) -> (mean, rstd); NaN when poisoned / overflowed
__device__ __forceinline__ void group_mean_rstd(const float *gt, int N, int n, int g, long HW, int cg, float eps, float &mean_f, float &rstd) {
    long long S = 0, Q = 0;
    bool bad = false;
    const long long *g0 = reinterpret_cast<const long long *>(gt) + ((long)n * 32 + g) * 2;
    const long pitch = (long)N * 64;                  // 64-bit words per shard
    // (the copies are requested TOGETHER: a loop of load - wait - add is eight round trips to L2 at the head of every workgroup of the upper levels)
    if (stat_shards(HW) == 8) {
        longlong2 t[8];
#pragma unroll
        for (int sh = 0; sh < 8; ++sh) t[sh] = *reinterpret_cast<const longlong2 *>(g0 + sh * pitch);
#pragma unroll
        for (int sh = 0; sh < 8; ++sh) {
            S += t[sh].x;
            Q += t[sh].y;
            bad |= t[sh].y < 0 || (unsigned long long)t[sh].y >= STAT_POISON;
        }
    } else {
        static_assert(stat_shards(0) == 1 && stat_shards(1l << 40) == 8, "stat_shards is 1 or 8");
        const longlong2 t = *reinterpret_cast<const longlong2 *>(g0);
        S = t.x;
        Q = t.y;
        bad = t.y < 0 || (unsigned long long)t.y >= STAT_POISON;
    }
    bad |= Q < 0 || (unsigned long long)Q >= STAT_POISON;
    const double cnt = (double)HW * cg;
    const double mean = (double)S * (1.0 / STAT_SC_SUM) / cnt;
    double var = (double)Q * (1.0 / STAT_SC_SQ) / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    mean_f = (float)mean;
    rstd = bad ? __builtin_nanf("") : (float)(1.0 / sqrt(var + (double)eps));
}

// ---- power-of-two activation scales of the fp16x2 kernels (round 6) -------------------------------------------------------------------------
// The two fp16 planes of an activation hold 2^-21 of it only while both are normal: 2^-3 <= |x| < 65504.  Below, the absolute error is 2^-25 - harmless
// when the tensor's rms is O(1), 2^-15 of the rms for a tensor at 2^-10; above, the round-toward-zero conversion saturates without a trace.  So the
// staging multiplies by a power of two sx chosen from a BOUND of the image's largest magnitude (exact; undone in the epilogue):
//   * raw input: |x| <= sqrt(sum x^2), the sum taken from the group totals the tensor's producer(s) left (act_scale_totals);
//   * behind a fused GroupNorm: |gamma' xhat + beta'| <= max|gamma'| sqrt(n_g) + max|beta'| (Cauchy-Schwarz over the group's n_g values; coef_to_lds;
//     gamma' = gamma (1 + scale), beta' = beta (1 + scale) + shift: a bound that needs the parameters and the embedding only).
// |x sx| <= 32752, so a plane cannot overflow, and whatever the tensor's magnitude the planes keep their precision relative to that bound.
__device__ __forceinline__ float pow2_scale_for_bound(float bound) {   // the power of two sx with bound * sx in (16376, 32752]; 1 when the bound is 0 / not finite
    if (!(bound > 0.f) || !(bound < 3.0e38f)) return 1.f;
    const float r = 32752.f / bound;
    unsigned e = __float_as_uint(r) >> 23;                           // (r > 0: no sign bit)
    e = e < 27u ? 27u : (e > 227u ? 227u : e);                       // 2^-100 ... 2^100
    return __uint_as_float(e << 23);
}
// every lane of the wave returns the same value; gt = [shard][N][32][2] totals of image n's tensor (any grouping), HW its pixels per image
__device__ __forceinline__ float act_scale_totals(const float *gt, int N, int n, long HW) {
    const int g = threadIdx.x & 31;
    long long Qi = 0;
    int bad = 0;
    const long long *g0 = reinterpret_cast<const long long *>(gt) + ((long)n * 32 + g) * 2 + 1;
    const long pitch = (long)N * 64;
    if (stat_shards(HW) == 8) {                       // (requested together, as in group_mean_rstd)
        long long q[8];
#pragma unroll
        for (int sh = 0; sh < 8; ++sh) q[sh] = g0[sh * pitch];
#pragma unroll
        for (int sh = 0; sh < 8; ++sh) {
            bad |= q[sh] < 0 || (unsigned long long)q[sh] >= STAT_POISON;
            Qi += q[sh];
        }
    } else {
        Qi = g0[0];
        bad = Qi < 0 || (unsigned long long)Qi >= STAT_POISON;
    }
    bad |= Qi < 0 || (unsigned long long)Qi >= STAT_POISON;
    // the 32 groups' sums of squares as floats (a bound needs no more; the 1.0001 below covers the rounding), added inside each row of 16 lanes with
    // four DPP steps in a fixed order, then the two rows (lanes 32-63 hold the same 32 values again)
    float Q = (float)Qi;
    Q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, Q), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    Q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, Q), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    Q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, Q), 0x141, 0xF, 0xF, true));    // row_half_mirror
    Q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, Q), 0x140, 0xF, 0xF, true));    // row_mirror
    const float Qt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Q), 0)) +
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Q), 16));
    // (a poisoned / overflowed block says nothing about the magnitudes: scale 1, and the planes' own range decides)
    return __ballot(bad) ? 1.f : pow2_scale_for_bound(sqrtf(Qt * (float)(1.0 / STAT_SC_SQ)) * 1.0001f + 1e-30f);
}

// sum over the wave's 64 lanes, the same in every lane's return, in a fixed order (four DPP steps inside each row of 16, then the four rows)
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));    // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));    // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// max over the wave's 64 lanes of an unsigned value, the same in every lane's return (four DPP steps leave each row of 16 with its maximum -
// max is idempotent, so the mirror steps may count a lane twice - then one readlane per row)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); v = v > t ? v : t;     // quad_perm [1,0,3,2]
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); v = v > t ? v : t;     // quad_perm [2,3,0,1]
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); v = v > t ? v : t;    // row_half_mirror
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true); v = v > t ? v : t;    // row_mirror
    const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16), r2 = __builtin_amdgcn_readlane(v, 32),
                   r3 = __builtin_amdgcn_readlane(v, 48);
    const unsigned a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}

// ---- consumer side: the affine of image n into LDS (sA[c], sB[c], c < C) ------------------------------------------------------------------
// From the arrays (cA != null), or from the view's group totals + the norm's parameters (GnSrc): 32 threads read their group's pair(s) and
// form (mean, rstd), every channel gets its (A, B) - the arithmetic of k_gn_coef_tot, bit for bit.  `scr` = COEF_SCR_FLOATS floats of LDS.  Call
// with all threads; ends with a barrier.  SCALE: the table is multiplied by the power of two sx that bounds the normalised tensor (above) and sx
// is returned (1 with arrays: nothing is known about the tensor they normalise); the caller's activation and epilogue account for it.
template <bool SCALE = false>
__device__ __forceinline__ float coef_to_lds(const float *cA, const float *cB, const GnSrc &gn, int N, int n, float *sA, float *sB, float *scr,
                                             int tid, int nthr) {
    const int C = gn.C;
    if (cA) {
        for (int c = tid; c < C; c += nthr) { sA[c] = cA[(long)n * C + c]; sB[c] = cB[(long)n * C + c]; }
        __syncthreads();
        return 1.f;
    }
    const int cg = C / 32;
    // The parameters of the thread's first NB channels (tid + k nthr) are requested together with the group totals - one round trip for everything a table of up to
    // NB nthr channels needs - and stay in registers for both passes below; only wider tables (the 1536-channel decoder inputs at 256 threads) load again, a batch
    // at a time.  (A loop of load - wait - use per channel, which is what the plain form compiles to, costs a round trip to L2 per iteration, twice with SCALE.)
    constexpr int NB = 4;
    float ga[NB], be[NB], sc[NB], sf[NB];
    auto load_batch = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int c = c0 + k * nthr < C ? c0 + k * nthr : C - 1;         // (clamped: unconditional loads, the stores below are predicated)
            ga[k] = gn.gamma[c]; be[k] = gn.beta[c];
            if (gn.emb) { sc[k] = gn.emb[(long)n * gn.emb_pitch + c]; sf[k] = gn.emb[(long)n * gn.emb_pitch + C + c]; }
        }
    };
    load_batch(tid);
    if (tid < 32) {
        float m, r;
        group_mean_rstd(gn.gt, N, n, tid, gn.HW, cg, gn.eps, m, r);
        scr[tid] = m; scr[32 + tid] = r;
    }
    const int nw = nthr >> 6;
    if (SCALE) {
        // max |gamma'|, max |beta'| of the image depend on the parameters and the embedding only, not on the statistics: formed while the 32
        // threads above wait for their totals, reduced inside the wave with DPP (non-negative floats order like their bit patterns) and left
        // per wave in scr[64 + w], scr[72 + w] - no atomics, no barrier and no pass over the table beyond what the unscaled form has.
        float gmax = 0.f, bmax = 0.f;
        for (int c0 = tid; c0 < C; c0 += NB * nthr) {
            if (c0 != tid) load_batch(c0);
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                float g1 = ga[k], b1 = be[k];
                if (gn.emb) { const float s1 = 1.f + sc[k]; g1 = g1 * s1; b1 = b1 * s1 + sf[k]; }
                gmax = fmaxf(gmax, fabsf(g1)); bmax = fmaxf(bmax, fabsf(b1));      // (a clamped slot repeats channel C - 1: harmless in a maximum)
            }
        }
        const unsigned gm = wave_max_u32(__float_as_uint(gmax)), bm = wave_max_u32(__float_as_uint(bmax));
        if ((tid & 63) == 0) { scr[64 + (tid >> 6)] = __uint_as_float(gm); scr[72 + (tid >> 6)] = __uint_as_float(bm); }
        if (C > NB * nthr) load_batch(tid);                                        // (the registers hold the last batch: fetch the first again)
    }
    __syncthreads();
    float sx = 1.f;
    if (SCALE) {
        float gmax = 0.f, bmax = 0.f;
        for (int w = 0; w < nw; ++w) { gmax = fmaxf(gmax, scr[64 + w]); bmax = fmaxf(bmax, scr[72 + w]); }
        sx = pow2_scale_for_bound(gmax * sqrtf((float)gn.HW * (float)cg) * 1.0001f + bmax);
    }
    for (int c0 = tid; c0 < C; c0 += NB * nthr) {
        if (c0 != tid) load_batch(c0);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int c = c0 + k * nthr;
            if (c < C) {
                const int g = c / cg;
                float a = scr[32 + g] * ga[k];
                float b = be[k] - scr[g] * a;
                if (gn.emb) {
                    const float s1 = 1.f + sc[k];
                    a = a * s1;
                    b = b * s1 + sf[k];
                }
                sA[c] = a * sx;        // (sx is a power of two: the scaled table is the exact table, scaled)
                sB[c] = b * sx;
            }
        }
    }
    __syncthreads();
    return sx;
}
constexpr int COEF_SCR_FLOATS = 80;   // 32 means, 32 rstd, 8 + 8 per-wave maxima

}  // namespace hl
