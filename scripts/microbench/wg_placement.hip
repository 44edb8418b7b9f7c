// Microbenchmark: which CU does workgroup blockIdx land on?  256 threads, 78 KB of LDS (two workgroups per CU), grid 3072, every
// workgroup spins for ~30 us so that the first 512 are co-resident.   hipcc --offload-arch=gfx950 -O3 wg_placement.hip -o wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void k(unsigned *out, long long spin) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 4 + 0] = hw; out[blockIdx.x * 4 + 1] = xcc;
        const long long t0 = wall_clock64();
        out[blockIdx.x * 4 + 2] = (unsigned)(t0 & 0xffffffff);
        while (wall_clock64() - t0 < spin) {}
        lds[0] = 1.f;
    }
    __syncthreads();
    if (lds[0] == 3.f) out[0] = 7;
}
int main() {
    const int G = 3072; unsigned *d; hipMalloc(&d, G * 16); hipMemset(d, 0, G * 16);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 78336, 0, d, 3000LL);   // 30 us at 100 MHz
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 4); hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
    // HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ...
    std::map<unsigned, std::vector<int>> cu;
    unsigned tmin = ~0u; for (int b = 0; b < G; ++b) tmin = h[b * 4 + 2] < tmin ? h[b * 4 + 2] : tmin;
    for (int b = 0; b < 1024; ++b) {
        const unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 0xf;
        const unsigned key = (xcc << 16) | (hw & 0xff00);
        if (h[b * 4 + 2] - tmin < 1000) cu[key].push_back(b);
    }
    printf("CUs seen in the first round: %zu\n", cu.size());
    int shown = 0;
    for (auto &kv : cu) { if (shown++ < 24) { printf("xcc %u hw %04x :", kv.first >> 16, kv.first & 0xffff); for (int b : kv.second) printf(" %d", b); printf("\n"); } }
    // distribution of (second - first) block index on a CU within the same xcd slot numbering (blockIdx >> 3)
    std::map<int, int> diff;
    for (auto &kv : cu) if (kv.second.size() >= 2) diff[(kv.second[1] >> 3) - (kv.second[0] >> 3)]++;
    for (auto &kv : diff) printf("slot distance %d: %d CUs\n", kv.first, kv.second);
    return 0;
}
