// Microbenchmark (round 4): why does the operand preparation of k_march_b3w (softplus + exact three-way bf16 split of accumulator tiles) not hide behind its
// MFMAs, when independent VALU instructions do (mfma_valu_two_waves.hip)?  Eight waves per workgroup (two per SIMD).  Per iteration a wave issues 48
// v_mfma_f32_32x32x16_bf16 into four tiles of accumulator set OUT and, one step behind every MFMA, prepares B operands from accumulator set IN with the kernel's
// own instruction sequences (b3 split: v_and / v_perm / v_sub; softplus: v_exp / v_log / fma / v_max_i32).  OUT and IN swap every iteration (layer ping-pong).
//   MODE 0: MFMAs only            MODE 1: + raw split of 32 values per 24 MFMAs (the kernel's ratio), sources = IN tiles (written by the previous iteration's MFMAs)
//   MODE 2: + softplus + split    MODE 3: as 2, the prepared operand IS the B operand of the following MFMAs (true dependency, as in the kernel)
//   MODE 4: as 2 but the sources are registers no MFMA ever writes
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 mfma_split_chain.hip -o mfma_split_chain && ./mfma_split_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Tmp { float x, y, ex, ey; unsigned hx, hy; };
__device__ __forceinline__ float max0(float x) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }
__device__ __forceinline__ unsigned hi(float x) { return __builtin_bit_cast(unsigned, x) & 0xffff0000u; }
__device__ __forceinline__ unsigned pack(float x, float y) { return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, y), __builtin_bit_cast(unsigned, x), 0x07060302u); }
// step T (0..5 with softplus, 0..2 raw) of pair Q of half H
template <int SP, int H, int Q, int T>
__device__ __forceinline__ void prep(const f32x16 &src, u32x4 (&pl)[3], Tmp &t) {
    constexpr int r = 8 * H + 2 * Q;
    if constexpr (SP && T == 0) { t.x = src[r]; t.y = src[r + 1]; t.ex = -1.44269504f * fabsf(t.x); t.ey = -1.44269504f * fabsf(t.y); }
    else if constexpr (SP && T == 1) { t.ex = __builtin_amdgcn_exp2f(t.ex); t.ey = __builtin_amdgcn_exp2f(t.ey); t.ex = 1.f + t.ex; }
    else if constexpr (SP && T == 2) { t.ey = 1.f + t.ey; t.ex = __builtin_amdgcn_logf(t.ex); t.ey = __builtin_amdgcn_logf(t.ey); }
    else if constexpr (SP && T == 3) { t.x = fmaf(0.693147181f, t.ex, max0(t.x)); t.y = fmaf(0.693147181f, t.ey, max0(t.y)); t.hx = hi(t.x); }
    else if constexpr (!SP && T == 0) { t.x = src[r]; t.y = src[r + 1]; t.hx = hi(t.x); }
    else if constexpr (T == (SP ? 4 : 1)) { t.hy = hi(t.y); pl[0][Q] = pack(t.x, t.y); t.x -= __builtin_bit_cast(float, t.hx); t.y -= __builtin_bit_cast(float, t.hy); t.hx = hi(t.x); }
    else { t.hy = hi(t.y); pl[1][Q] = pack(t.x, t.y); t.x -= __builtin_bit_cast(float, t.hx); t.y -= __builtin_bit_cast(float, t.hy); pl[2][Q] = pack(t.x, t.y); }
}
template <int N, class F, int... I> __device__ __forceinline__ void seq_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void seq(F &&f) { seq_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// MODE 5 / 6: as MODE 3, and the A operands are weight fragments read from LDS the way k_march_b3w reads them: six ds_read_b128 (two tiles x three planes) in
// front of every 12 MFMAs (MODE 5), or the same reads issued one group EARLY (MODE 6: the reads of group g + 1 behind the first MFMAs of group g, 24 more registers)
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float *src, int iters, float *sink) {
    __shared__ __attribute__((aligned(16))) u32x4 frag[3072];     // 48 KB: two chunk pairs of fragments
    for (int i = threadIdx.x; i < 3072; i += 512) frag[i] = u32x4{0x3f803f80u, (unsigned)i, 0, 0x3f80u};
    __syncthreads();
    f32x16 A[4], B[4], Cst[4];
    const float a0 = src[threadIdx.x];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) { A[t][r] = a0 * (r + 1); B[t][r] = a0 * (r + 2); Cst[t][r] = a0 * (r + 3) + 0.5f; }
    u32x4 w = {__builtin_bit_cast(unsigned, a0), 0x3f803f80u, 0, 0x3f80u};
    u32x4 pa[3], pb[3];
    for (int p = 0; p < 3; ++p) { pa[p] = w; pb[p] = w; }
    Tmp tm{};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    const int lane = threadIdx.x & 63;
    u32x4 wf[2][3], wn[2][3];
    int gcount = 0;
    auto rd = [&](u32x4 (&d)[2][3], int g) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) d[t][p] = frag[((g & 7) * 6 + t * 3 + p) * 64 + lane];
    };
    if constexpr (MODE == 6) rd(wn, 0);
    auto layer = [&](f32x16 (&IN)[4], f32x16 (&OUT)[4]) {
        if constexpr (MODE >= 5) {
            constexpr int PW[6] = {2, 1, 0, 1, 0, 0};
            seq<4>([&](auto gc) {                                   // four groups of 12 MFMAs (two tiles x six products)
                constexpr int g = decltype(gc)::value, c = g >> 1;
                u32x4 (&use)[3] = (c == 0) ? pa : pb;
                u32x4 (&mk)[3] = (c == 0) ? pb : pa;
                if constexpr (MODE == 5) rd(wf, gcount + g);
                else {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int p = 0; p < 3; ++p) wf[t][p] = wn[t][p];
                }
                seq<12>([&](auto ic) {
                    constexpr int idx = decltype(ic)::value, t = idx & 1, i = idx >> 1;
                    OUT[2 * (g & 1) + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[t][PW[i]]), __builtin_bit_cast(bf16x8, use[PB[i]]), OUT[2 * (g & 1) + t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (MODE == 6 && idx == 1) rd(wn, gcount + g + 1);
                    prep<1, c, ((g & 1) * 12 + idx) / 6, ((g & 1) * 12 + idx) % 6>(IN[c], mk, tm);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            gcount += 4;
            return;
        }
        // 48 MFMAs = two "chunks" of 24 (4 tiles x 6 products); behind chunk c the operand of the other kind is prepared from IN[c]
        seq<2>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            u32x4 (&use)[3] = (c == 0) ? pa : pb;
            u32x4 (&mk)[3] = (c == 0) ? pb : pa;
            seq<24>([&](auto ic) {
                constexpr int idx = decltype(ic)::value, t = idx & 3, i = idx >> 2;
                const u32x4 &bop = (MODE == 3) ? use[PB[i]] : w;
                OUT[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, bop), OUT[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE == 1) { if constexpr (idx < 12) prep<0, c, idx / 3, idx % 3>(IN[c], mk, tm); }
                else if constexpr (MODE == 2 || MODE == 3) prep<1, c, idx / 6, idx % 6>(IN[c], mk, tm);
                else if constexpr (MODE == 4) prep<1, c, idx / 6, idx % 6>(Cst[c], mk, tm);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
#ifndef UNROLL_PAIRS
#define UNROLL_PAIRS 1      // layer pairs (192 MFMAs + their preparation, ~1 100 instructions = ~8 KB of code) per loop iteration: the code size knob
#endif
    for (int it = 0; it < iters; it += UNROLL_PAIRS) {
#pragma unroll
        for (int u = 0; u < UNROLL_PAIRS; ++u) {
            layer(A, B);
            layer(B, A);
        }
    }
    float o = 0.f;
    for (int t = 0; t < 4; ++t) o += A[t][0] + B[t][3] + Cst[t][1];
    for (int p = 0; p < 3; ++p) o += __builtin_bit_cast(float, (pa[p][0] ^ pb[p][1]) & 0x007fffffu);
    sink[blockIdx.x * 512 + threadIdx.x] = o + tm.x;
}

template <int MODE>
static void run(const char *what, const float *src, float *sink) {
    const int iters = 960, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 512>>>(src, 10, sink);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 512>>>(src, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-100s %7.3f ms  %6.1f cycles per MFMA and SIMD (nominal 2.4 GHz; 2 waves x 96 MFMAs per iteration)\n", what, ms, ms * 1e-3 * 2.4e9 / iters / 192.0);
}

int main() {
    float *src, *sink;
    (void)hipMalloc(&src, 1 << 20); (void)hipMalloc(&sink, 256 * 512 * 4);
    (void)hipMemset(src, 0, 1 << 20);
    run<0>("MFMAs only", src, sink);
    run<1>("+ raw split of the input tiles (12 steps per 24 MFMAs)", src, sink);
    run<2>("+ softplus + split of the input tiles (24 steps per 24 MFMAs), operands not consumed", src, sink);
    run<3>("+ softplus + split, the prepared planes ARE the next chunk's B operands", src, sink);
    run<4>("+ softplus + split of registers no MFMA writes", src, sink);
    run<5>("+ softplus + split (as 3) + six fragment ds_read_b128 in front of every 12 MFMAs", src, sink);
    run<6>("+ softplus + split (as 3) + the fragment reads issued one group early", src, sink);
    return 0;
}
