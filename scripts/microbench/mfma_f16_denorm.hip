#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* o) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f16v c; for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) o[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float vals[][2] = {{1.f, 1.f}, {9.5367431640625e-07f /*2^-20*/, 1024.f}, {-9.5367431640625e-07f, 1024.f}, {3.0517578125e-05f /*2^-15*/, 1.f}, {6.103515625e-05f /*2^-14*/, 1.f}, {5.9604644775390625e-08f /*2^-24*/, 4096.f}};
    for (auto& v : vals) {
        k<<<1, 64>>>(v[0], v[1], d); float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g: mfma sum of 16 products = %g (expected %g)\n", v[0], v[1], h, 16.0 * v[0] * v[1]);
    }
}
