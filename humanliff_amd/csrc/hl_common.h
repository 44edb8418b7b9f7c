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
// The empty asm pins x and y as the fp32 values they are.  Without it the backend folds a producing multiply into the conversion for the SCALAR uses of h0
// (v_fma_mixlo_f16 rounds the exact product once) while the packed plane still takes v_cvt_pk_f16_f32 of the rounded product; where the two roundings differ -
// one fp16 ulp, a few values in a thousand - the residual is formed against another h0 than the one stored, and the pair is off by 2^-11 (found in the attention's
// scaled K operand: max-abs 4e-5 instead of 2e-6; -ffp-contract=off does not stop this fold).
__device__ __forceinline__ void hl_split2_rne(float x, float y, unsigned &p0, unsigned &p1) {
    asm("" : "+v"(x), "+v"(y));
    typedef _Float16 hl_h2 __attribute__((ext_vector_type(2)));
    const hl_h2 h0 = {(_Float16)x, (_Float16)y};
    const hl_h2 h1 = {(_Float16)(x - (float)h0[0]), (_Float16)(y - (float)h0[1])};
    p0 = __builtin_bit_cast(unsigned, h0);
    p1 = __builtin_bit_cast(unsigned, h1);
}
#endif
