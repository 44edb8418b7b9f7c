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
