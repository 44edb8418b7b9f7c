#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 att_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void att_split8(const f32x4 a, const f32x4 b, float sc, u32x4 &p0, u32x4 &p1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x = (q < 2 ? a[2 * q] : b[2 * q - 4]) * sc, y = (q < 2 ? a[2 * q + 1] : b[2 * q - 3]) * sc;
        const att_f16x2 h0 = {(_Float16)x, (_Float16)y}, h1 = {(_Float16)(x - (float)h0[0]), (_Float16)(y - (float)h0[1])};
        p0[q] = __builtin_bit_cast(unsigned, h0);
        p1[q] = __builtin_bit_cast(unsigned, h1);
    }
}
__global__ void k(const float* x, float sc, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (8 * i + 7 >= n) return;
    f32x4 a = *(const f32x4*)(x + 8 * i), b = *(const f32x4*)(x + 8 * i + 4);
    u32x4 p0, p1;
    att_split8(a, b, sc, p0, p1);
    for (int q = 0; q < 4; ++q) {
        att_f16x2 h0 = __builtin_bit_cast(att_f16x2, p0[q]), h1 = __builtin_bit_cast(att_f16x2, p1[q]);
        o[8 * i + 2 * q] = (float)h0[0] + (float)h1[0];
        o[8 * i + 2 * q + 1] = (float)h0[1] + (float)h1[1];
    }
}
int main() {
    const int n = 1 << 16;
    float *hx = new float[n], *ho = new float[n];
    for (int i = 0; i < n; ++i) { float e = -30.f + 36.f * (i / (float)n); hx[i] = ldexpf(1.f + (i % 977) / 977.f, (int)e) * ((i & 1) ? -1.f : 1.f); }
    float *dx, *dout; hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    for (float sc : {1.f, 0.3195f}) {
        k<<<n / 8 / 256, 256>>>(dx, sc, dout, n);
        hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
        printf("sc = %g\n", sc);
        for (int i = 40000; i < 40016; ++i) printf("    x %.9g -> %.9g\n", hx[i] * sc, ho[i]);
        for (int eb = -30; eb < 6; eb += 4) {
            double worst = 0;
            for (int i = 0; i < n; ++i) { float xs = hx[i] * sc; int e; frexpf(xs, &e); if (e - 1 >= eb && e - 1 < eb + 4) { double rel = fabs((double)ho[i] - xs) / fabs(xs); if (rel > worst) worst = rel; } }
            printf("  2^%d..2^%d: worst rel err %.3e\n", eb, eb + 4, worst);
        }
    }
    return 0;
}
