// Microbenchmark (round 4): the instruction mix of a Winograd k-tile on the 16-bit matrix pipe.  ONE wave per SIMD issues a stream of
// v_mfma_f32_32x32x16_{bf16,f16} on 16 accumulator tiles with, behind every MFMA, NV plain VALU instructions (v_fma_f32 / v_and_b32 /
// v_sub_f32 / v_perm_b32 in rotation, as the transform + the three-way bf16 split would issue them), optionally one ds_read_b128 and, every
// LD_EVERY-th gap, one buffer_load_dwordx4 of a 1 KiB weight fragment from an 8 MB weight image that every CU walks in step (what the real
// kernel asks of the L2).  Cycles per MFMA by s_memtime and by wall clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 mfma16_mix.hip -o mfma16_mix && ./mfma16_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int T, int F16>
__device__ __forceinline__ void mfma_lit(u32x4 a, u32x4 b) {
#define M(N, LO, HI, ...)                                                                                                                  \
    if constexpr (T == N) {                                                                                                                \
        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 a[" #LO ":" #HI "], %0, %1, a[" #LO ":" #HI "]" ::"v"(a), "v"(b) : __VA_ARGS__); \
        else asm volatile("v_mfma_f32_32x32x16_bf16 a[" #LO ":" #HI "], %0, %1, a[" #LO ":" #HI "]" ::"v"(a), "v"(b) : __VA_ARGS__);     \
    }
    M(0, 0, 15, "a0", "a15") M(1, 16, 31, "a16", "a31") M(2, 32, 47, "a32", "a47") M(3, 48, 63, "a48", "a63") M(4, 64, 79, "a64", "a79") M(5, 80, 95, "a80", "a95")
    M(6, 96, 111, "a96", "a111") M(7, 112, 127, "a112", "a127") M(8, 128, 143, "a128", "a143") M(9, 144, 159, "a144", "a159") M(10, 160, 175, "a160", "a175")
    M(11, 176, 191, "a176", "a191") M(12, 192, 207, "a192", "a207") M(13, 208, 223, "a208", "a223") M(14, 224, 239, "a224", "a239") M(15, 240, 255, "a240", "a255")
#undef M
}

// NV: VALU per gap; LDS: ds_read_b128 per gap; LD_EVERY: one weight load every LD_EVERY gaps (0 = none); F16: f16 instead of bf16 MFMA
template <int NV, int LDS, int LD_EVERY, int F16, int IND>
__global__ __launch_bounds__(256, 1) void k(const float *src, const float *wimg, int iters, float *sink, long long *cyc, int span, int rotate) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)wimg, (short)0, 8 << 20, 0x00020000);
    asm volatile(".set mf_i, 0\n\t.rept 256\n\tv_accvgpr_write_b32 a[mf_i], 0\n\t.set mf_i, mf_i+1\n\t.endr" ::: "a0", "a255");
    const float a = src[threadIdx.x], b = src[threadIdx.x + 256];
    u32x4 A = {__builtin_bit_cast(unsigned, a), 0x3f803f80u, 0, 0x3f80u}, B = {__builtin_bit_cast(unsigned, b), 0x3f80u, 0, 0};
    float s[16];
    unsigned m[8];
    f32x4 d[4];
    u32x4 g[12];
    for (int i = 0; i < 16; ++i) s[i] = a * i;
    for (int i = 0; i < 8; ++i) m[i] = __builtin_bit_cast(unsigned, b) + i;
    for (int i = 0; i < 4; ++i) d[i] = {0, 0, 0, 0};
    for (int i = 0; i < 12; ++i) g[i] = {0, 0, 0, 0};
    const float c1 = 0.999f;
    const unsigned sel = 0x07060302u, msk = 0xffff0000u;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[i];
    __syncthreads();
    // weight walk: n-block = blockIdx % 3 owns a third of the image; every workgroup of an n-block reads the same fragments in the same order
    const int nb = blockIdx.x % 3;
    const int wbase = nb * span;                           // span: bytes of one n-block's weight image (a multiple of 36 KB)
    int wpos = rotate ? (int)((blockIdx.x / 3 * 7u) % (unsigned)(span / (36 << 10))) * (36 << 10) : 0;
    int woff = wbase + wpos + wave * (9 << 10);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        [&]<int... J>(std::integer_sequence<int, J...>) {
            ([&] {
                constexpr int tile = ((J >> 3) * 2 + (J & 1)) % 16;
                mfma_lit<tile, F16>(A, B);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NV; ++f) {
                    constexpr int dummy = 0;
                    const int q = J * NV + f;
                    if constexpr (IND == 2) { asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s[q & 15]) : "v"(m[q & 7]), "v"(m[(q + 3) & 7])); continue; }
                    if constexpr (IND == 3) { asm volatile("v_exp_f32 %0, %1" : "=v"(s[q & 15]) : "v"(s[(q + 5) & 15])); continue; }
                    if constexpr (IND == 4) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m[q & 7]) : "v"(s[(q + 5) & 15]), "v"(s[(q + 9) & 15])); continue; }
                    if constexpr (IND == 5) { asm volatile("v_accvgpr_read_b32 %0, a[7]" : "=v"(s[q & 15])); continue; }
                    if constexpr (IND == 6) { asm volatile("v_max_i32 %0, 0, %1" : "=v"(m[q & 7]) : "v"(m[(q + 3) & 7])); continue; }
                    switch (IND ? 0 : (q & 3)) {
                        case 0: asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[q & 15]) : "v"(a), "s"(c1)); break;
                        case 1: asm volatile("v_and_b32 %0, %1, %2" : "=v"(m[q & 7]) : "s"(msk), "v"(s[(q + 5) & 15])); break;
                        case 2: asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s[q & 15]) : "v"(s[(q + 3) & 15]), "v"(m[(q + 2) & 7])); break;
                        default: asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(m[q & 7]) : "v"(s[(q + 7) & 15]), "v"(s[(q + 9) & 15]), "s"(sel)); break;
                    }
                }
#pragma unroll
                for (int f = 0; f < LDS; ++f) d[(J + f) & 3] = *reinterpret_cast<const f32x4 *>(lds + lane * 4 + ((J + f) & 7) * 256 + wave * 2048);
                if constexpr (LD_EVERY > 0) {
                    if constexpr (J % LD_EVERY == 0) {
                        constexpr int slot = (J / LD_EVERY) % 12;
                        A[2] ^= g[slot][J & 3];             // consume the fragment loaded 12 loads ago (a counted vmcnt wait), then refill its slot
                        g[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, woff + (J / LD_EVERY) * 1024, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 108>{});
        wpos += 36 << 10;                                  // next k-tile (the image wraps)
        if (wpos + (144 << 10) > span) wpos = 0;
        woff = wbase + wpos + wave * (9 << 10);
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");
    float r0;
    asm volatile("v_accvgpr_read_b32 %0, a[0]" : "=v"(r0));
    float acc = r0;
    for (int i = 0; i < 16; ++i) acc += s[i];
    for (int i = 0; i < 8; ++i) acc += (float)m[i];
    for (int i = 0; i < 4; ++i) acc += d[i][0] + d[i][3];
    for (int i = 0; i < 12; ++i) acc += __builtin_bit_cast(f32x4, g[i])[1];
    if (sink) sink[blockIdx.x * 256 + threadIdx.x] = acc;
    if (cyc && lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NV, int LDS, int LD_EVERY, int F16 = 0, int IND = 0>
void run(const float *src, const float *wimg, float *sink, long long *cyc, int span = 2592 << 10, int rotate = 0) {
    const int iters = 100, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, LDS, LD_EVERY, F16, IND>), dim3(grid), dim3(256), 65536, 0, src, wimg, 10, sink, cyc, span, rotate);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, LDS, LD_EVERY, F16, IND>), dim3(grid), dim3(256), 65536, 0, src, wimg, iters, sink, cyc, span, rotate);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 1024; ++i) avg += (double)h[i]; avg /= 1024;
    const double per = avg / (iters * 108.0), ns_per = ms * 1e6 / (iters * 108.0);
    printf("[span %4d KB rot %d] %s%s VALU/gap=%2d ds_read/gap=%d weight load every %d gaps : %6.2f ns per MFMA, counter %6.2f cycles per MFMA -> %5.0f cycles per 108-MFMA k-tile, eff clock %.2f GHz\n",
           span >> 10, rotate, F16 ? "f16 " : "bf16", IND == 1 ? " (independent v_fma)" : IND == 2 ? " (v_dot2c_f32_bf16)" : IND == 3 ? " (v_exp_f32)" : IND == 4 ? " (v_cvt_pk_bf16_f32)" : IND == 5 ? " (v_accvgpr_read)" : IND == 6 ? " (v_max_i32)" : "", NV, LDS, LD_EVERY, ns_per, per, per * 108.0, per / ns_per);
}

// do f16 MFMAs honour subnormal inputs?  A = 2^-20 (f16 subnormal) in every k slot of row m, B = 1: D = 16 * 2^-20 if honoured, 0 if flushed
__global__ void k_sub(float *out) {
    u32x4 A = {0x00100010u, 0x00100010u, 0x00100010u, 0x00100010u}, B = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A), __builtin_bit_cast(h8, B), c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    u32x4 Ab = {0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u}, Bb = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};   // bf16 subnormal 2^-133
    f32x16 e = {};
    e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, Ab), __builtin_bit_cast(b8, Bb), e, 0, 0, 0);
    if (threadIdx.x == 0) out[1] = e[0] * 0x1p100f;
}

int main() {
    float *src, *wimg, *sink; long long *cyc;
    hipMalloc(&src, 1 << 20); hipMalloc(&wimg, 8 << 20); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    hipMemset(src, 0, 1 << 20); hipMemset(wimg, 0, 8 << 20);
    k_sub<<<1, 64>>>(sink);
    float hs[2]; hipMemcpy(hs, sink, 8, hipMemcpyDeviceToHost);
    printf("f16 MFMA, subnormal inputs 16 x 2^-20 x 1: %g (honoured: %g)   bf16 subnormal 16 x 2^-133 x 1 (x 2^100): %g (honoured: %g)\n", hs[0], 16 * 0x1p-20, hs[1], 16 * 0x1p-33);
    run<0, 0, 0>(src, wimg, sink, cyc);
    run<2, 0, 0>(src, wimg, sink, cyc); run<4, 0, 0>(src, wimg, sink, cyc); run<5, 0, 0>(src, wimg, sink, cyc); run<6, 0, 0>(src, wimg, sink, cyc);
    run<7, 0, 0>(src, wimg, sink, cyc); run<8, 0, 0>(src, wimg, sink, cyc); run<10, 0, 0>(src, wimg, sink, cyc); run<12, 0, 0>(src, wimg, sink, cyc);
    run<0, 1, 0>(src, wimg, sink, cyc); run<0, 0, 2>(src, wimg, sink, cyc); run<0, 0, 1>(src, wimg, sink, cyc);
    run<4, 1, 2>(src, wimg, sink, cyc); run<6, 1, 2>(src, wimg, sink, cyc); run<7, 1, 2>(src, wimg, sink, cyc); run<8, 1, 2>(src, wimg, sink, cyc);
    run<10, 1, 2>(src, wimg, sink, cyc);
    run<4, 0, 0, 0, 2>(src, wimg, sink, cyc); run<8, 0, 0, 0, 2>(src, wimg, sink, cyc); run<12, 0, 0, 0, 2>(src, wimg, sink, cyc);
    run<2, 0, 0, 0, 3>(src, wimg, sink, cyc); run<4, 0, 0, 0, 3>(src, wimg, sink, cyc); run<8, 0, 0, 0, 3>(src, wimg, sink, cyc);
    run<8, 0, 0, 0, 4>(src, wimg, sink, cyc); run<12, 0, 0, 0, 4>(src, wimg, sink, cyc);
    run<8, 0, 0, 0, 5>(src, wimg, sink, cyc); run<12, 0, 0, 0, 5>(src, wimg, sink, cyc);
    run<8, 0, 0, 0, 6>(src, wimg, sink, cyc); run<12, 0, 0, 0, 6>(src, wimg, sink, cyc);
    run<6, 0, 0, 0, 1>(src, wimg, sink, cyc); run<8, 0, 0, 0, 1>(src, wimg, sink, cyc); run<12, 0, 0, 0, 1>(src, wimg, sink, cyc);
    // the f16 two-plane variant: 54 MFMAs per k-tile carry the same transform and a cheaper split -> ~10 VALU per gap, a weight load every 1.5 gaps
    run<0, 0, 0, 1>(src, wimg, sink, cyc); run<8, 1, 2, 1>(src, wimg, sink, cyc); run<10, 1, 1, 1>(src, wimg, sink, cyc); run<12, 1, 1, 1>(src, wimg, sink, cyc);
    return 0;
}
