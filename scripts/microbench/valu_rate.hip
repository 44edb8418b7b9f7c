// Issue rate of the instructions the fp16x2 splits are made of (gfx950): one wave per SIMD, REP independent instructions in an unrolled loop, cycles from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define BODY(INS) \
    _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) { \
        asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(y)); }
#define I_PKRTZ(i) "v_cvt_pkrtz_f16_f32 %" #i ", %8, %9\n"
#define I_PKRNE(i) "v_cvt_pk_f16_f32 %" #i ", %8, %9\n"
#define I_MIX(i) "v_fma_mix_f32 %" #i ", %8, -1.0, %9 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"
#define I_CVT32(i) "v_cvt_f32_f16_e32 %" #i ", %8\n"
#define I_CVT16(i) "v_cvt_f16_f32_e32 %" #i ", %8\n"
#define I_AND(i) "v_and_b32_e32 %" #i ", %8, %9\n"
#define I_SUB(i) "v_sub_f32_e32 %" #i ", %8, %9\n"
#define I_PKMUL(i) "v_pk_mul_f32 %" #i ", %10, %10\n"
#define I_FMA(i) "v_fma_f32 %" #i ", %8, %9, %9\n"
#define I_EXP(i) "v_exp_f32_e32 %" #i ", %8\n"
#define I_MED3(i) "v_med3_f32 %" #i ", %8, %9, %9\n"
template <int K>
__global__ __launch_bounds__(256) void k(float *out, float x, float y, int iters) {
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a2[8];
    for (int i = 0; i < 8; ++i) a2[i] = f2{0, 0};
    const f2 x2 = {x, y};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (K == 0) BODY(I_PKRTZ)
        if constexpr (K == 1) BODY(I_PKRNE)
        if constexpr (K == 2) BODY(I_MIX)
        if constexpr (K == 3) BODY(I_CVT32)
        if constexpr (K == 4) BODY(I_CVT16)
        if constexpr (K == 5) BODY(I_AND)
        if constexpr (K == 6) BODY(I_SUB)
        if constexpr (K == 7) {
#pragma unroll
            for (int r = 0; r < REP / 8; ++r)
                asm volatile("v_pk_mul_f32 %0, %8, %8\nv_pk_mul_f32 %1, %8, %8\nv_pk_mul_f32 %2, %8, %8\nv_pk_mul_f32 %3, %8, %8\nv_pk_mul_f32 %4, %8, %8\nv_pk_mul_f32 %5, %8, %8\nv_pk_mul_f32 %6, %8, %8\nv_pk_mul_f32 %7, %8, %8\n"
                             : "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "+v"(a2[6]), "+v"(a2[7]) : "v"(x2));
        }
        if constexpr (K == 8) BODY(I_FMA)
        if constexpr (K == 9) BODY(I_EXP)
        if constexpr (K == 10) BODY(I_MED3)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + a2[i][0];
    if (s == 12345.f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0) / ((float)iters * REP);
}
template <int K>
void run(const char *name, float *d) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<K><<<256, 256>>>(d, 1.5f, 2.5f, 100);
    hipEventRecord(e0);
    k<K><<<256, 256>>>(d, 1.5f, 2.5f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 1 wave per SIMD: cycles per instruction per wave = ms * clock / (iters * REP); report ns per instruction and relative to v_and
    printf("%-22s %.3f ns per instruction per wave (one wave per SIMD)\n", name, ms * 1e6 / ((double)iters * REP));
}
int main() {
    float *d; hipMalloc(&d, 64);
    run<5>("v_and_b32", d); run<6>("v_sub_f32", d); run<8>("v_fma_f32", d); run<7>("v_pk_mul_f32", d); run<0>("v_cvt_pkrtz_f16_f32", d); run<1>("v_cvt_pk_f16_f32", d);
    run<2>("v_fma_mix_f32", d); run<3>("v_cvt_f32_f16", d); run<4>("v_cvt_f16_f32", d); run<10>("v_med3_f32", d); run<9>("v_exp_f32", d);
    return 0;
}
