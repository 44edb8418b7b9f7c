// Does the rate of v_mfma_f32_32x32x16_f16 depend on the operand VALUES?  (It does: the chip runs under a power cap, and operands with more significant bits
// switch more of the multiplier array.)  Four waves per CU x 2 (one or two per SIMD), a long chain of independent MFMAs on register operands that are
//   zeros | fp16 subnormals with 3 significant bits (what the low plane of an unscaled |w| ~ 0.02 weight looks like) | normal fp16 with full mantissas (scaled planes)
// Prints TFLOP/s of each over ~0.5 s, after a warm-up that brings the chip to its steady clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(const unsigned short *a, const unsigned short *b, float *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    h8 A[4], B[4];
    for (int i = 0; i < 4; ++i) { A[i] = *(const h8 *)(a + ((t * 4 + i) & 65535) * 8); B[i] = *(const h8 *)(b + ((t * 4 + i) & 65535) * 8); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[j], B[(i + j) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 1234.5f) out[0] = s;
}
int main() {
    const int n = 65536 * 8;
    unsigned short *h = new unsigned short[n];
    unsigned short *da, *db; float *dout;
    hipMalloc(&da, n * 2); hipMalloc(&db, n * 2); hipMalloc(&dout, 64);
    const char *names[] = {"zeros", "subnormal, 3 significant bits", "normal, 5 significant bits", "normal, full mantissa"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
        srand(7);
        for (int i = 0; i < n; ++i) {
            unsigned short v = 0;
            const unsigned r = (unsigned)rand();
            if (mode == 1) v = (unsigned short)(((r & 7u) << 4) | ((r >> 8) & 1u) << 15);                          // exponent 0, mantissa bits 4..6
            if (mode == 2) v = (unsigned short)((((r >> 4) % 6 + 12) << 10) | ((r & 0xfu) << 6) | ((r >> 16) & 1u) << 15);
            if (mode == 3) v = (unsigned short)((((r >> 10) % 6 + 12) << 10) | (r & 0x3ffu) | ((r >> 16) & 1u) << 15);
            h[i] = v;
        }
        hipMemcpy(da, h, n * 2, hipMemcpyHostToDevice); hipMemcpy(db, h, n * 2, hipMemcpyHostToDevice);
        const int iters = 40000, grid = 512;
        k<<<grid, 256>>>(da, db, dout, iters);                  // warm-up at this operand pattern
        hipEventRecord(e0);
        k<<<grid, 256>>>(da, db, dout, iters);
        k<<<grid, 256>>>(da, db, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * grid * 4 * (double)iters * 16 * 2.0 * 32 * 32 * 16;
        printf("pass %d  %-32s %8.1f TFLOP/s (%.1f ms)\n", rep, names[mode], flop / (ms * 1e-3) * 1e-12, ms);
    }
    return 0;
}
