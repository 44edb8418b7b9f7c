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
// consumers turn that (and an overflowed total) into NaN coefficients.
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
    const int ns = stat_shards(HW);
    for (int sh = 0; sh < ns; ++sh) {
        const longlong2 t = *reinterpret_cast<const longlong2 *>(reinterpret_cast<const long long *>(gt) + (((long)sh * N + n) * 32 + g) * 2);
        S += t.x;
        Q += t.y;
        bad |= t.y < 0 || (unsigned long long)t.y >= STAT_POISON;
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
//   * behind a fused GroupNorm: |gamma' xhat + beta'| <= max|gamma'| sqrt(n_g) + max|beta'| (Cauchy-Schwarz over the group's n_g values; coef_to_lds).
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
    double Q = 0.0;
    int bad = 0;
    const int ns = stat_shards(HW);
    for (int sh = 0; sh < ns; ++sh) {
        const long long q = reinterpret_cast<const long long *>(gt)[(((long)sh * N + n) * 32 + g) * 2 + 1];
        bad |= q < 0 || (unsigned long long)q >= STAT_POISON;
        Q += (double)q;
    }
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        Q += __shfl_xor(Q, d);
        bad |= __shfl_xor(bad, d);
    }
    // (a poisoned / overflowed block says nothing about the magnitudes: scale 1, and the planes' own range decides)
    return bad ? 1.f : pow2_scale_for_bound((float)sqrt(Q * (1.0 / STAT_SC_SQ)) * 1.0001f + 1e-30f);
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
    // (the parameters of the thread's first channel are requested together with the group totals: one round trip instead of two)
    float ga0 = 0.f, be0 = 0.f, sc0 = 0.f, sf0 = 0.f;
    if (tid < C) {
        ga0 = gn.gamma[tid]; be0 = gn.beta[tid];
        if (gn.emb) { sc0 = gn.emb[(long)n * gn.emb_pitch + tid]; sf0 = gn.emb[(long)n * gn.emb_pitch + C + tid]; }
    }
    if (tid < 32) {
        float m, r;
        group_mean_rstd(gn.gt, N, n, tid, gn.HW, cg, gn.eps, m, r);
        scr[tid] = m; scr[32 + tid] = r;
    }
    if (SCALE && tid == 0) { scr[64] = 0.f; scr[65] = 0.f; }
    __syncthreads();
    float gmax = 0.f, bmax = 0.f;                      // max |gamma'|, max |beta'| over this thread's channels
    for (int c = tid; c < C; c += nthr) {
        const int g = c / cg;
        const bool first = c == tid;
        float ga = first ? ga0 : gn.gamma[c], be = first ? be0 : gn.beta[c];
        float a = scr[32 + g] * ga;
        float b = be - scr[g] * a;
        if (gn.emb) {
            const float sc = 1.f + (first ? sc0 : gn.emb[(long)n * gn.emb_pitch + c]);
            const float sf = first ? sf0 : gn.emb[(long)n * gn.emb_pitch + C + c];
            a = a * sc;
            b = b * sc + sf;
            ga = ga * sc; be = be * sc + sf;
        }
        sA[c] = a;
        sB[c] = b;
        if (SCALE) { gmax = fmaxf(gmax, fabsf(ga)); bmax = fmaxf(bmax, fabsf(be)); }
    }
    if (!SCALE) { __syncthreads(); return 1.f; }
    // (non-negative floats order like their bit patterns)
    atomicMax(reinterpret_cast<unsigned *>(scr + 64), __float_as_uint(gmax));
    atomicMax(reinterpret_cast<unsigned *>(scr + 65), __float_as_uint(bmax));
    __syncthreads();
    const float sx = pow2_scale_for_bound(scr[64] * sqrtf((float)gn.HW * (float)cg) * 1.0001f + scr[65]);
    if (sx != 1.f) {
        for (int c = tid; c < C; c += nthr) { sA[c] *= sx; sB[c] *= sx; }
    }
    __syncthreads();
    return sx;
}
constexpr int COEF_SCR_FLOATS = 68;

}  // namespace hl
