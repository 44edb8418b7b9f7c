// Shared host-side helpers for libhumanliff_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/humanliff_hip.h"

namespace hl {

inline char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(hl_status st, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return (int)st;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(HL_ERR_RUNTIME, "%s: %s", what, hipGetErrorString(e));
    return HL_OK;
}

}  // namespace hl

#define HL_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return hl::fail(HL_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define HL_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) return hl::fail(HL_ERR_RUNTIME, #call ": %s", hipGetErrorString(e__)); \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#if defined(__HIPCC__)
// Two fp32 values -> the two packed fp16 planes of the fp16x2 products (round 6): h0 = the NEAREST fp16 (v_cvt_pk_f16_f32), h1 = the nearest fp16 of the residual
// against h0 converted back (exact in fp32): |x - h0 - h1| <= 2^-23 |x| while both planes are normal; a value beyond fp16's range becomes inf / NaN (loud).
// Four instructions per pair, written out: the residuals come from v_fma_mix_f32, which reads one fp16 half of the packed h0 as an operand (x - h0 = fma(h0, -1, x),
// exact), so h0 is never unpacked.  Left to the compiler the same C costs eight (it converts every value twice, once packed for the plane and once alone for the
// residual) - and worse, it folds a producing multiply into the SCALAR conversion only (v_fma_mixlo_f16 rounds the exact product once) while the packed plane still
// takes v_cvt_pk_f16_f32 of the rounded product: where the two roundings differ - one fp16 ulp, a few values in a thousand - the residual is formed against another
// h0 than the one stored and the pair is off by 2^-11 (found in the attention's scaled K operand: max-abs 4e-5 instead of 2e-6; -ffp-contract=off does not stop
// that fold).  scripts/microbench/split_mix.hip checks these four instructions against the plain C form bit for bit, subnormals, infinities and NaN included.
// the same planes left to the compiler (v_cvt_pk_f16_f32, two unpacking conversions, v_pk_fma_f32 / two subtractions, v_cvt_pk_f16_f32): for kernels whose register
// allocation the asm form upsets.  The empty asm statements pin the operands, see above.
__device__ __forceinline__ void hl_split2_rne_c(float x, float y, unsigned &p0, unsigned &p1) {
    typedef _Float16 hl_h2 __attribute__((ext_vector_type(2)));
    typedef float hl_f2 __attribute__((ext_vector_type(2)));
    asm("" : "+v"(x), "+v"(y));
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector((hl_f2){x, y}, hl_h2));
    asm("" : "+v"(p0));
    const hl_h2 h0 = __builtin_bit_cast(hl_h2, p0);
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector((hl_f2){x - (float)h0[0], y - (float)h0[1]}, hl_h2));
}
__device__ __forceinline__ void hl_split2_rne(float x, float y, unsigned &p0, unsigned &p1) {
    // (ONE asm statement: the compiler, which treats every asm statement as a hazard it cannot see into, pads once per pair instead of four times; x and y are
    // read only - the callers' values often stay live - and the residuals pass through p1 and one scratch register)
    float t;
    asm("v_cvt_pk_f16_f32 %0, %3, %4\n\t"
        "v_fma_mix_f32 %1, %0, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %1, %1, %2"
        : "=&v"(p0), "=&v"(p1), "=&v"(t) : "v"(x), "v"(y));
}
#endif
