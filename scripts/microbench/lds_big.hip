// Does a 512-thread workgroup get (and keep) dynamic LDS beyond 100 KB on gfx950?  Writes a pattern at every 16-byte slot of a 132 KB allocation, reads it back after barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void k(int nbytes, unsigned* bad) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int n = nbytes / 4;
    for (int r = 0; r < 4; ++r) {
        for (int i = threadIdx.x; i < n; i += 512) lds[i] = (unsigned)i * 2654435761u + r + blockIdx.x;
        __syncthreads();
        unsigned b = 0;
        for (int i = threadIdx.x; i < n; i += 512) b += lds[(i + 4096 * (r + 1)) % n] != (unsigned)((i + 4096 * (r + 1)) % n) * 2654435761u + r + blockIdx.x;
        if (b) atomicAdd(bad, b);
        __syncthreads();
    }
}
int main() {
    unsigned* d; hipMalloc(&d, 4);
    for (int kb : {64, 100, 124, 132, 160}) {
        hipMemset(d, 0, 4);
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        k<<<1024, 512, kb * 1024>>>(kb * 1024, d);
        hipError_t e2 = hipDeviceSynchronize();
        unsigned h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("%d KB: setattr %s, run %s, mismatches %u\n", kb, hipGetErrorString(e), hipGetErrorString(e2), h);
    }
}
