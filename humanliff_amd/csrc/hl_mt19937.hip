// sample_pdf's uniforms, bit for bit, on the device (human_diffusion/NeRF/renderer.py:545: `u = torch.rand(...)` on the CPU generator,
// then `.to(device)`).
//
// The reference draws the inverse-CDF uniforms with the process-wide CPU generator: mt19937 (ATen/core/MT19937RNGEngine.h), one 32-bit word
// per float, u = (word & (2^24 - 1)) * 2^-24 (ATen/core/TransformationHelper.h uniform_real<float>).  Drawn on the host, the 33.5 M numbers of
// a 512 x 512 view at n_importance = 128 cost 50 - 80 ms of one CPU core against 40 ms of GPU work for the whole view.  This kernel continues
// the SAME stream from the generator's current state (uploaded: 624 words + position) and hands back the state behind the last number, so the
// host generator is advanced as if it had drawn them: a drop-in call with u = None sees the reference's numbers.
//
// mt19937 is a linear recurrence x[k+624] = x[k+397] ^ twist(x[k], x[k+1]): 227 consecutive words are independent of each other, so one
// regeneration of the 624-word block is three dependent phases (words 0..226, 227..453, 454..623).  Thread t forms words t, t + 227, t + 454:
// the lag-397 operand of a phase is the word the same thread formed in the phase before, so a block needs ONE barrier (two LDS images, the
// previous block is only read), each new word tempered, masked and stored as it is formed.  One workgroup of 256 threads; the launch runs
// on its own stream next to the evaluate passes (the importance-sampling launch alone waits for it, hl_render_rays_u_event).
#include "hl_common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr unsigned MT_A = 0x9908b0dfu, MT_UP = 0x80000000u, MT_LO = 0x7fffffffu;

__device__ __forceinline__ unsigned mt_twist(unsigned u, unsigned v) { return (((u & MT_UP) | (v & MT_LO)) >> 1) ^ ((v & 1u) ? MT_A : 0u); }
__device__ __forceinline__ float mt_uniform(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return (float)(y & 0xffffffu) * 5.9604644775390625e-08f;   // (24 bits: exact in fp32) * 2^-24
}

// st_in: 624 state words; pos: words of the current block already handed out (624: the block is used up - also the freshly seeded generator);
// out[0..n): the next n uniforms of the stream; st_out: 624 words + the new position.
__global__ __launch_bounds__(256) void k_mt19937_uniform(const unsigned *__restrict__ st_in, int pos, float *__restrict__ out, long n,
                                                         unsigned *__restrict__ st_out) {
    __shared__ unsigned buf[2][MT_N];
    const int tid = threadIdx.x;
    for (int i = tid; i < MT_N; i += 256) buf[0][i] = st_in[i];
    __syncthreads();
    int cur = 0;
    long done = 0;
    if (pos < MT_N) {   // the rest of the block the host generator was in
        const long cnt = min(n, (long)(MT_N - pos));
        for (int i = tid; i < cnt; i += 256) out[i] = mt_uniform(buf[0][pos + i]);
        done = cnt;
        pos += (int)cnt;
    }
    // One regeneration per barrier: thread t < 227 forms words t, t + 227 and t + 454 - the lag-397 operand of the second and third is the word
    // the SAME thread has just formed (x[k + 397 - 624] = x[k - 227]), so only values of the previous block are read from LDS (all reads of a
    // thread are issued together: one LDS latency per block); word 623 twists with the NEW word 0, which its thread forms again for itself.
    const int k2 = tid + (MT_N - MT_M), k3 = tid + 2 * (MT_N - MT_M);
    const bool act = tid < MT_N - MT_M, has3 = k3 < MT_N;
    while (done < n) {
        const unsigned *o = buf[cur];
        unsigned *w = buf[cur ^ 1];
        const long left = n - done;
        if (act) {
            const unsigned a0 = o[tid], a1 = o[tid + 1], am = o[tid + MT_M], b0 = o[k2], b1 = o[k2 + 1];
            const unsigned c0 = has3 ? o[k3] : 0u, c1o = (has3 && k3 + 1 < MT_N) ? o[k3 + 1] : 0u;
            const unsigned n0 = o[MT_M] ^ mt_twist(o[0], o[1]);          // the new word 0 (broadcast reads)
            const unsigned y1 = am ^ mt_twist(a0, a1);
            const unsigned y2 = y1 ^ mt_twist(b0, b1);
            w[tid] = y1;
            w[k2] = y2;
            if (tid < left) out[done + tid] = mt_uniform(y1);
            if (k2 < left) out[done + k2] = mt_uniform(y2);
            if (has3) {
                const unsigned y3 = y2 ^ mt_twist(c0, k3 == MT_N - 1 ? n0 : c1o);
                w[k3] = y3;
                if (k3 < left) out[done + k3] = mt_uniform(y3);
            }
        }
        // (LDS only: __syncthreads() would also wait for the global stores of this block - their latency, not the recurrence, then sets the pace)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur ^= 1;
        pos = (int)min(left, (long)MT_N);
        done += pos;
    }
    for (int i = tid; i < MT_N; i += 256) st_out[i] = buf[cur][i];
    if (tid == 0) st_out[MT_N] = (unsigned)pos;
}

}  // namespace

extern "C" int hl_mt19937_uniform(const uint32_t *state, int pos, float *out, int64_t n, uint32_t *state_out, void *stream) {
    HL_REQUIRE(state && out && state_out && n >= 0 && pos >= 0 && pos <= MT_N, "hl_mt19937_uniform: bad argument");
    hipLaunchKernelGGL(k_mt19937_uniform, dim3(1), dim3(256), 0, (hipStream_t)stream, state, pos, out, (long)n, state_out);
    return hl::check_launch("k_mt19937_uniform");
}
