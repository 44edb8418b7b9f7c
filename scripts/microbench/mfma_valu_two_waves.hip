// Microbenchmark (round 4): do the MFMAs of one wave and the vector-ALU instructions of ANOTHER wave on the same SIMD overlap?
// A workgroup of 8 waves = two per SIMD.  Waves 0-3 ("M") issue NM back-to-back v_mfma_f32_32x32x16_bf16 on four independent accumulator tiles
// per iteration; waves 4-7 ("V") issue NV plain VALU instructions (v_fma_f32 / v_and_b32 / v_sub_f32 / v_perm_b32 in rotation, independent
// chains) per iteration.  Modes: M only (V waves exit), V only, both.  If the SIMD overlaps the two pipes across waves, "both" takes
// max(M, V); if it serialises them, the sum.  Also "same": every wave issues NM MFMAs + NV VALU per iteration (what k_march_b3w's waves do).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 mfma_valu_two_waves.hip -o mfma_valu_two_waves && ./mfma_valu_two_waves
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NM>
__device__ __forceinline__ void mfmas(f32x16 (&acc)[4], bf16x8 a, bf16x8 b) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
}
template <int NV>
__device__ __forceinline__ void valus(float (&s)[8], unsigned (&m)[8]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = i & 7;
#ifdef TRANS_EVERY      // every TRANS_EVERY-th instruction is a transcendental (v_exp_f32 / v_log_f32 alternating): k_march_b3w issues 576 among 4 320
        if (i % TRANS_EVERY == TRANS_EVERY - 1) {
            if ((i / TRANS_EVERY) & 1) asm volatile("v_log_f32 %0, %0" : "+v"(s[k]));
            else asm volatile("v_exp_f32 %0, %0" : "+v"(s[k]));
            continue;
        }
#endif
        if ((i & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(s[k]) : "v"(s[(k + 1) & 7]));
        else if ((i & 3) == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(m[k]));
        else if ((i & 3) == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s[k]) : "v"(s[(k + 3) & 7]));
        else asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(m[k]) : "v"(m[(k + 1) & 7]), "v"(0x07060302u));
    }
}

// MODE 0: M waves only; 1: V waves only; 2: both (separate waves); 3: every wave does both, MFMAs first then VALU; 4: every wave, interleaved 1 MFMA : NV/NM VALU
template <int NM, int NV, int MODE>
__global__ __launch_bounds__(512, 1) void k(const float *src, int iters, float *sink, long long *cyc) {
    const int wave = threadIdx.x >> 6;
    const bool isM = wave < 4;
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float s[8];
    unsigned m[8];
    const float a0 = src[threadIdx.x];
    for (int i = 0; i < 8; ++i) { s[i] = a0 * (i + 1) * 1e-3f; m[i] = __builtin_bit_cast(unsigned, a0) + i; }
    u32x4 au = {__builtin_bit_cast(unsigned, a0), 0x3f803f80u, 0, 0x3f80u}, bu = {0x3f80u, __builtin_bit_cast(unsigned, a0), 0, 0};
    const bf16x8 A = __builtin_bit_cast(bf16x8, au), B = __builtin_bit_cast(bf16x8, bu);
    if ((MODE == 0 && !isM) || (MODE == 1 && isM)) return;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) mfmas<NM>(acc, A, B);
        else if constexpr (MODE == 1) valus<NV>(s, m);
        else if constexpr (MODE == 2) { if (isM) mfmas<NM>(acc, A, B); else valus<NV>(s, m); }
        else if constexpr (MODE == 3) { mfmas<NM / 2>(acc, A, B); __builtin_amdgcn_sched_barrier(0); valus<NV / 2>(s, m); __builtin_amdgcn_sched_barrier(0); }
        else {
#pragma unroll
            for (int q = 0; q < NM / 2; ++q) {
                acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[q & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                valus<NV / NM>(s, m);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float o = 0.f;
    for (int t = 0; t < 4; ++t) o += acc[t][0];
    for (int i = 0; i < 8; ++i) o += s[i] + __builtin_bit_cast(float, m[i] & 0x007fffffu);
    sink[blockIdx.x * 512 + threadIdx.x] = o;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NM, int NV, int MODE>
static void run(const char *what, const float *src, float *sink, long long *cyc) {
    const int iters = 2000, blocks = 256;
    hipMemset(cyc, 0, blocks * 8 * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NM, NV, MODE><<<blocks, 512>>>(src, 10, sink, cyc);
    hipEventRecord(e0);
    k<NM, NV, MODE><<<blocks, 512>>>(src, iters, sink, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (long long v : h) mx = v > mx ? v : mx;
    // s_memtime ticks at 100 MHz on this part: use the wall clock and a nominal 2.4 GHz for cycles
    const double cyc_it = ms * 1e-3 * 2.4e9 / iters;
    printf("NM=%3d NV=%3d %-44s %8.3f ms  %8.1f cycles/iteration (at 2.4 GHz)  [per SIMD: %d MFMA x 32 = %d, %d VALU x 4 = %d]\n", NM, NV, what, ms, cyc_it,
           MODE >= 3 ? NM : NM, NM * 32, NV, NV * 4);
}

int main() {
    float *src, *sink; long long *cyc;
    hipMalloc(&src, 1 << 20); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    hipMemset(src, 0, 1 << 20);
#define SET(NM, NV)                                                                                   \
    run<NM, NV, 0>("M waves only (4 waves, one per SIMD)", src, sink, cyc);                          \
    run<NM, NV, 1>("V waves only", src, sink, cyc);                                                  \
    run<NM, NV, 2>("M waves + V waves (two per SIMD)", src, sink, cyc);                              \
    run<NM, NV, 3>("8 waves, each: NM/2 MFMA then NV/2 VALU", src, sink, cyc);                       \
    run<NM, NV, 4>("8 waves, each: 1 MFMA : NV/NM VALU interleaved", src, sink, cyc);
#ifdef TRANS_EVERY
    printf("one VALU instruction in %d is v_exp_f32 / v_log_f32\n", TRANS_EVERY);
#endif
    SET(48, 192)    // 4 VALU per MFMA
    SET(48, 288)    // 6 VALU per MFMA (k_march_b3w: 5.6)
    SET(48, 384)    // 8 VALU per MFMA
    return 0;
}
