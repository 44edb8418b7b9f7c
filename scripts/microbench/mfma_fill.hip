// Microbenchmark: what does ONE wave per SIMD hide in the 64-cycle shadow of a v_mfma_f32_32x32x2_f32?  A stream of MFMAs on 16 accumulator
// tiles (the accumulator file, like k_conv_wino4w) with NF filler instructions of one kind behind every MFMA; cycles per MFMA by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 mfma_fill.hip -o mfma_fill && ./mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int T>
__device__ __forceinline__ void mfma_lit(float a, float b) {
#define M(N, LO, HI, ...) if constexpr (T == N) asm volatile("v_mfma_f32_32x32x2_f32 a[" #LO ":" #HI "], %0, %1, a[" #LO ":" #HI "]" ::"v"(a), "v"(b) : __VA_ARGS__);
    M(0, 0, 15, "a0", "a15") M(1, 16, 31, "a16", "a31") M(2, 32, 47, "a32", "a47") M(3, 48, 63, "a48", "a63") M(4, 64, 79, "a64", "a79") M(5, 80, 95, "a80", "a95")
    M(6, 96, 111, "a96", "a111") M(7, 112, 127, "a112", "a127") M(8, 128, 143, "a128", "a143") M(9, 144, 159, "a144", "a159") M(10, 160, 175, "a160", "a175")
    M(11, 176, 191, "a176", "a191") M(12, 192, 207, "a192", "a207") M(13, 208, 223, "a208", "a223") M(14, 224, 239, "a224", "a239") M(15, 240, 255, "a240", "a255")
#undef M
}
// KIND 0 none, 1 v_pk_fma_f32 (f32x2 fma), 2 v_fma_f32 (asm), 3 ds_read_b128, 4 buffer_load_dwordx4 -> VGPR (L2-hot), 5 LDS-DMA 1 KiB piece,
// 6 v_fma_f32 (compiler), 7 s_add (SALU), 8 v_pk_fma with a VGPR constant (no SGPR operand), 9 v_exp_f32 (transcendental), 10 v_mov_b32
template <int KIND, int NF, int ORDER>
__global__ __launch_bounds__(256, 1) void k(const float *src, int iters, float *sink, long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, (short)0, 1 << 20, 0x00020000);
    asm volatile(".set mf_i, 0\n\t.rept 256\n\tv_accvgpr_write_b32 a[mf_i], 0\n\t.set mf_i, mf_i+1\n\t.endr" ::: "a0", "a255");
    float a = src[threadIdx.x], b = src[threadIdx.x + 256];
    f32x2 p[8], q[8];
    float s[16];
    f32x4 d[8];
    u32x4 g[8];
    for (int i = 0; i < 8; ++i) { p[i] = {a + i, b + i}; q[i] = {b - i, a - i}; d[i] = {0, 0, 0, 0}; g[i] = {0, 0, 0, 0}; }
    for (int i = 0; i < 16; ++i) s[i] = a * i;
    const float c1 = 0.999f;
    int sacc = iters;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[i];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        [&]<int... J>(std::integer_sequence<int, J...>) {
            ([&] {
                // ORDER 0: consecutive MFMAs on different tiles (pairs interleaved like the kernel); 1: four MFMAs in a row on the same tile
                constexpr int tile = ORDER == 0 ? ((J >> 3) * 2 + (J & 1)) % 16 : (J >> 2) % 16;
                mfma_lit<tile>(a, b);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    constexpr int dummy = 0;
                    const int r = (J * NF + f) & 7;
                    if constexpr (KIND == 1) p[r] = __builtin_elementwise_fma((f32x2)(c1), p[r], q[r]);
                    else if constexpr (KIND == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[(J * NF + f) & 15]) : "v"(a), "s"(c1));
                    else if constexpr (KIND == 3) d[r] = *reinterpret_cast<const f32x4 *>(lds + lane * 4 + r * 256 + wave * 2048);
                    else if constexpr (KIND == 4) g[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, (r * 1024 + wave * 8192) & 0xfffff, 0);
                    else if constexpr (KIND == 5) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + wave * 2048 + r * 256), 16,
                                                                                          lane * 16u, (r * 1024 + wave * 8192) & 0xfffff, 0, 0);
                    else if constexpr (KIND == 6) s[(J * NF + f) & 15] = __builtin_fmaf(s[(J * NF + f) & 15], c1, a);
                    else if constexpr (KIND == 7) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
                    else if constexpr (KIND == 8) p[r] = __builtin_elementwise_fma(q[(r + 1) & 7], p[r], q[r]);
                    else if constexpr (KIND == 9) s[(J * NF + f) & 15] = __builtin_amdgcn_exp2f(s[(J * NF + f) & 15]);
                    else if constexpr (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(s[(J * NF + f) & 15]) : "v"(a));
                }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 72>{});
        if constexpr (KIND == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");
    float r0;
    asm volatile("v_accvgpr_read_b32 %0, a[0]" : "=v"(r0));
    float acc = r0 + sacc;
    for (int i = 0; i < 8; ++i) acc += p[i][0] + p[i][1] + d[i][0] + d[i][3] + __builtin_bit_cast(f32x4, g[i])[1];
    for (int i = 0; i < 16; ++i) acc += s[i];
    if (sink) sink[blockIdx.x * 256 + threadIdx.x] = acc;
    if (cyc && lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int KIND, int NF, int ORDER = 0>
void run(const char *name, const float *src, float *sink, long long *cyc) {
    const int iters = 200, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NF, ORDER>), dim3(grid), dim3(256), 65536, 0, src, 20, sink, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NF, ORDER>), dim3(grid), dim3(256), 65536, 0, src, iters, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 1024; ++i) avg += (double)h[i]; avg /= 1024;
    const double per = avg / (iters * 72.0);   // s_memtime ticks (100 MHz constant clock on this part?) - report wall-derived cycles too
    const double ns_per = ms * 1e6 / (iters * 72.0);
    printf("%-34s NF=%d order=%d : %7.2f ns per MFMA (%.1f cyc @2.4GHz; counter %.2f ticks)  -> extra %.1f ns per filler\n", name, NF, ORDER, ns_per, ns_per * 2.4, per,
           NF ? (ns_per - 26.7) / NF : 0.0);
}

int main() {
    float *src, *sink; long long *cyc;
    hipMalloc(&src, 1 << 20); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    hipMemset(src, 0, 1 << 20);
    run<0, 0>("none", src, sink, cyc);
    run<0, 0, 1>("none", src, sink, cyc);
    run<1, 1>("v_pk_fma_f32 (sgpr const)", src, sink, cyc); run<1, 2>("v_pk_fma_f32 (sgpr const)", src, sink, cyc); run<1, 4>("v_pk_fma_f32 (sgpr const)", src, sink, cyc); run<1, 8>("v_pk_fma_f32 (sgpr const)", src, sink, cyc);
    run<8, 1>("v_pk_fma_f32 (vgpr)", src, sink, cyc); run<8, 2>("v_pk_fma_f32 (vgpr)", src, sink, cyc); run<8, 4>("v_pk_fma_f32 (vgpr)", src, sink, cyc);
    run<2, 1>("v_fma_f32 asm", src, sink, cyc); run<2, 2>("v_fma_f32 asm", src, sink, cyc); run<2, 4>("v_fma_f32 asm", src, sink, cyc); run<2, 8>("v_fma_f32 asm", src, sink, cyc);
    run<6, 2>("v_fma_f32 compiler", src, sink, cyc); run<6, 4>("v_fma_f32 compiler", src, sink, cyc); run<6, 8>("v_fma_f32 compiler", src, sink, cyc);
    run<3, 1>("ds_read_b128", src, sink, cyc); run<3, 2>("ds_read_b128", src, sink, cyc); run<3, 4>("ds_read_b128", src, sink, cyc);
    run<4, 1>("buffer_load_dwordx4", src, sink, cyc); run<4, 2>("buffer_load_dwordx4", src, sink, cyc);
    run<5, 1>("LDS-DMA piece", src, sink, cyc);
    run<9, 1>("v_exp_f32", src, sink, cyc); run<9, 2>("v_exp_f32", src, sink, cyc); run<9, 4>("v_exp_f32", src, sink, cyc); run<9, 8>("v_exp_f32", src, sink, cyc);
    run<10, 4>("v_mov_b32", src, sink, cyc); run<10, 8>("v_mov_b32", src, sink, cyc);
    run<7, 2>("s_add_i32", src, sink, cyc); run<7, 8>("s_add_i32", src, sink, cyc);
    run<1, 2, 1>("v_pk_fma_f32, same-tile runs", src, sink, cyc); run<1, 4, 1>("v_pk_fma_f32, same-tile runs", src, sink, cyc);
    return 0;
}
