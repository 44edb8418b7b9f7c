// hl_split2_rne as four instructions (v_cvt_pk_f16_f32, two v_fma_mix_f32 reading one fp16 half each, v_cvt_pk_f16_f32) against the plain C form:
// bit-identical planes over random values, every exponent from 2^-40 to 2^17, fp16 subnormal range, +-0, inf, NaN.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include "../../humanliff_amd/csrc/hl_common.h"
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ref_split(float x, float y, unsigned &p0, unsigned &p1) {
    asm("" : "+v"(x), "+v"(y));
    const h2 h0 = {(_Float16)x, (_Float16)y};
    float rx = x - (float)h0[0], ry = y - (float)h0[1];
    asm("" : "+v"(rx), "+v"(ry));
    const h2 h1 = {(_Float16)rx, (_Float16)ry};
    p0 = __builtin_bit_cast(unsigned, h0);
    p1 = __builtin_bit_cast(unsigned, h1);
}
__global__ void k(const float *x, unsigned *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned a0, a1, b0, b1;
    hl_split2_rne(x[2 * i], x[2 * i + 1], a0, a1);
    ref_split(x[2 * i], x[2 * i + 1], b0, b1);
    o[4 * i] = a0; o[4 * i + 1] = a1; o[4 * i + 2] = b0; o[4 * i + 3] = b1;
}
int main() {
    const int n = 1 << 22;
    float *hx = new float[n];
    unsigned *ho = new unsigned[2 * n];
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int e = -40 + (i % 58);
        float m = 1.f + (rand() & 0x7fffff) / 8388608.f;
        hx[i] = ldexpf(m, e) * ((rand() & 1) ? -1.f : 1.f);
    }
    const float special[] = {0.f, -0.f, INFINITY, -INFINITY, NAN, 65504.f, 65519.9f, 65520.f, 1e30f, 6.1e-5f, 5.96e-8f, 2.98e-8f, 1e-45f, -65536.f};
    for (int i = 0; i < (int)(sizeof(special) / 4); ++i) hx[i] = special[i];
    float *dx; unsigned *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, 2 * n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(dx, dout, n);
    hipMemcpy(ho, dout, 2 * n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n / 2; ++i) {
        const bool same = ho[4 * i] == ho[4 * i + 2] && ho[4 * i + 1] == ho[4 * i + 3];
        // (two NaNs may differ in payload: compare as NaN)
        if (!same) {
            bool nan_ok = true;
            for (int h = 0; h < 2; ++h) for (int pl = 0; pl < 2; ++pl) {
                unsigned a = (ho[4 * i + pl] >> (16 * h)) & 0xffff, b = (ho[4 * i + 2 + pl] >> (16 * h)) & 0xffff;
                const bool an = (a & 0x7c00) == 0x7c00 && (a & 0x3ff), bn = (b & 0x7c00) == 0x7c00 && (b & 0x3ff);
                if (a != b && !(an && bn)) nan_ok = false;
            }
            if (!nan_ok) { if (bad < 10) printf("MISMATCH x = %.9g, %.9g: %08x %08x vs %08x %08x\n", hx[2 * i], hx[2 * i + 1], ho[4 * i], ho[4 * i + 1], ho[4 * i + 2], ho[4 * i + 3]); ++bad; }
        }
    }
    printf("split_mix: %d pairs, %ld mismatches\n", n / 2, bad);
    return bad != 0;
}
