// Cost of the split of EIGHT values into the two fp16 planes, as the conv staging issues it (one wave per SIMD, nothing to hide behind):
//   A  four asm statements of 4 dependent instructions (pair after pair)      B  one statement, the four pairs interleaved (dependent distance 4)
//   C  round 5's truncating split left to the compiler (v_and, v_sub, v_cvt_pkrtz)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split2(float x, float y, unsigned &p0, unsigned &p1) {
    float t;
    asm volatile("v_cvt_pk_f16_f32 %0, %3, %4\n\tv_fma_mix_f32 %1, %0, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_cvt_pk_f16_f32 %1, %1, %2"
        : "=&v"(p0), "=&v"(p1), "=&v"(t) : "v"(x), "v"(y));
}
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, u32x4 &p0, u32x4 &p1) {
    float t0, t1, t2, t3;
    unsigned a0, a1, a2, a3, b0, b1, b2, b3;
    asm volatile("v_cvt_pk_f16_f32 %0, %12, %13\n\tv_cvt_pk_f16_f32 %1, %14, %15\n\tv_cvt_pk_f16_f32 %2, %16, %17\n\tv_cvt_pk_f16_f32 %3, %18, %19\n\t"
        "v_fma_mix_f32 %4, %0, -1.0, %12 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, %1, -1.0, %14 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %6, %2, -1.0, %16 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %7, %3, -1.0, %18 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %8, %0, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %9, %1, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %10, %2, -1.0, %17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %11, %3, -1.0, %19 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %4, %4, %8\n\tv_cvt_pk_f16_f32 %5, %5, %9\n\tv_cvt_pk_f16_f32 %6, %6, %10\n\tv_cvt_pk_f16_f32 %7, %7, %11"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    p0 = u32x4{a0, a1, a2, a3}; p1 = u32x4{b0, b1, b2, b3};
}
__device__ __forceinline__ void split2_rtz(float x, float y, unsigned &p0, unsigned &p1) {
    const float hx = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffffe000u), hy = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, y) & 0xffffe000u);
    p0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(hx, hy));
    p1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x - hx, y - hy));
}
template <int K>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters) {
    f32x4 a = *(const f32x4 *)(in + threadIdx.x * 8), b = *(const f32x4 *)(in + threadIdx.x * 8 + 4);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned q0[4], q1[4];
        u32x4 p0, p1;
        if (K == 0) { split2(a[0], a[1], q0[0], q1[0]); split2(a[2], a[3], q0[1], q1[1]); split2(b[0], b[1], q0[2], q1[2]); split2(b[2], b[3], q0[3], q1[3]); p0 = u32x4{q0[0], q0[1], q0[2], q0[3]}; p1 = u32x4{q1[0], q1[1], q1[2], q1[3]}; }
        if (K == 1) split8(a, b, p0, p1);
        if (K == 2) { split2_rtz(a[0], a[1], q0[0], q1[0]); split2_rtz(a[2], a[3], q0[1], q1[1]); split2_rtz(b[0], b[1], q0[2], q1[2]); split2_rtz(b[2], b[3], q0[3], q1[3]); p0 = u32x4{q0[0], q0[1], q0[2], q0[3]}; p1 = u32x4{q1[0], q1[1], q1[2], q1[3]}; }
        acc ^= p0[0] ^ p0[1] ^ p0[2] ^ p0[3] ^ p1[0] ^ p1[1] ^ p1[2] ^ p1[3];
        asm volatile("" : "+v"(a), "+v"(b));      // (the values change as far as the compiler knows: nothing is hoisted)
    }
    if (acc == 0x12345u) out[0] = 1.f;
}
template <int K>
void run(const char *name, float *d, float *in) {
    const int iters = 100000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<K><<<256, 256>>>(d, in, 100);
    hipEventRecord(e0);
    k<K><<<256, 256>>>(d, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.2f ns per 8 values per wave\n", name, ms * 1e6 / iters);
}
int main() {
    float *d, *in; hipMalloc(&d, 64); hipMalloc(&in, 256 * 8 * 4); hipMemset(in, 0x3c, 256 * 8 * 4);
    run<0>("A: 4 statements of 4 (RNE, fma_mix)", d, in); run<1>("B: one statement, interleaved (RNE)", d, in); run<2>("C: truncating, compiler (round 5)", d, in);
    return 0;
}
