// Microbenchmark: how many bytes per cycle and CU does the LDS-DMA path (buffer_load_dwordx4 ... lds) deliver from an L2-resident
// buffer, against plain buffer_load_dwordx4 into registers (+ ds_write_b128)?   hipcc --offload-arch=gfx950 -O3 dma_bw.hip -o dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: LDS-DMA b128; 1: LDS-DMA b32 (4 instructions per 1 KiB... 256 B each); 2: buffer_load_dwordx4 to VGPR + ds_write_b128; 3: to VGPR only
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const float *src, int bytes_mask, int iters, float *sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, (short)0, bytes_mask + 1, 0x00020000);
    float *my = lds + wave * 8 * 256;
    unsigned base = (blockIdx.x * 7919u * 1024u + wave * 8192u) & bytes_mask;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned off = (base + j * 1024u) & bytes_mask;
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(my + j * 256), 16, lane * 16u, off, 0, 0);
            } else if (MODE >= 7) {
                // 1 KiB contiguous, but the 16-byte pieces permuted inside every 128-byte group of eight lanes (7: XOR 5, 8: XOR (group & 7)):
                // does the address coalescing depend on the lane order inside a group?
                const unsigned grp = lane >> 3, k = (lane & 7) ^ (MODE == 7 ? 5u : (grp & 7u));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(my + j * 256), 16, (grp * 8u + k) * 16u, off, 0, 0);
            } else if (MODE >= 4) {
                // gathers shaped like the convolution's patch DMA: 768-byte pixels (192 channels); 4: one 16-byte piece per lane and pixel,
                // 5: lanes 2i, 2i+1 take 32 contiguous bytes of a pixel, 6: lanes 4i..4i+3 two pixels' 32 bytes twice (nearest-x2 upsample)
                const unsigned pix = MODE == 4 ? lane : (MODE == 5 ? (lane >> 1) : (lane >> 2));
                const unsigned sub = MODE == 4 ? 0u : (lane & 1) * 16u;
                const unsigned voff = (pix * 768u + sub + ((unsigned)it & 3u) * 32u) & (unsigned)bytes_mask;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(my + j * 256), 16, voff, (off & ~1023u) * 48u & (unsigned)bytes_mask, 0, 0);
            } else if (MODE == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(my + j * 256 + q * 64), 4, lane * 4u, off + q * 256, 0, 0);
            } else {
                const f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, off, 0) ;
                if (MODE == 2) *reinterpret_cast<f32x4 *>(my + j * 256 + lane * 4) = v; else acc += v;
            }
        }
        base = (base + 65536u * WAVES) & bytes_mask;
        if (MODE <= 1 || MODE >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // keep one batch in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink) sink[threadIdx.x] = lds[threadIdx.x] + acc[0] + acc[1] + acc[2] + acc[3];
}
template <int MODE, int WAVES>
void run(const char *name, const float *src, int bytes, int wgs_per_cu, float *sink) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    const size_t shm = WAVES * 8 * 1024;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), shm, 0, src, bytes - 1, 50, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), shm, 0, src, bytes - 1, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double tot = (double)grid * WAVES * iters * 8 * 1024;
    printf("%-34s waves/WG %d WG/CU %d buffer %5d KB: %7.2f TB/s = %5.1f B/ns/CU\n", name, WAVES, wgs_per_cu, bytes >> 10, tot / ms * 1e-9, tot / ms * 1e-6 / 256);
}
int main() {
    float *src, *sink; const int maxb = 64 << 20;
    hipMalloc(&src, maxb); hipMemset(src, 0, maxb); hipMalloc(&sink, 4096);
    for (int bytes : {1 << 20, 16 << 20}) {
        run<0, 4>("LDS-DMA b128", src, bytes, 2, sink);
        run<0, 4>("LDS-DMA b128", src, bytes, 1, sink);
        run<0, 8>("LDS-DMA b128", src, bytes, 2, sink);
        run<1, 4>("LDS-DMA b32", src, bytes, 2, sink);
        run<4, 4>("LDS-DMA b128 gather 16 B / pixel", src, bytes, 2, sink);
        run<5, 4>("LDS-DMA b128 gather 32 B / pixel", src, bytes, 2, sink);
        run<6, 4>("LDS-DMA b128 gather 32 B / pixel x2", src, bytes, 2, sink);
        run<7, 4>("LDS-DMA b128 1 KiB, groups of 8 permuted (^5)", src, bytes, 2, sink);
        run<8, 4>("LDS-DMA b128 1 KiB, groups of 8 permuted (^g)", src, bytes, 2, sink);
        run<2, 4>("buffer_load b128 + ds_write_b128", src, bytes, 2, sink);
        run<3, 4>("buffer_load b128 to VGPR", src, bytes, 2, sink);
        run<3, 8>("buffer_load b128 to VGPR", src, bytes, 2, sink);
    }
    return 0;
}
