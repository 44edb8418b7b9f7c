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
// previous block is only read).  One workgroup of 256 threads walks the recurrence - the only sequential part - and stores the RAW state words
// (round 6; buffer stores with a scalar offset: no address arithmetic, and a word behind the end is dropped by the bounds check, so the loop
// has no predicates); a second, chip-wide launch tempers and converts them in place.  Rounds 4-5 tempered inside the walk: 20.7 ms per 33.5 M
// numbers, bound by its own vector instructions; the walk alone takes 14.5 ms (270 ns per block of 624: one LDS round trip and one barrier), and a
// drop-in render() call of a 512 x 512 view 30.5 ms against 33 before (26.1 - 26.4 with resident uniforms: `render.host_inclusive` 0.80 -> 0.87 of
// `render.value`).  The launches run on their own stream next to the evaluate passes - the draw for view k + 1 beside the fine pass of view k - and only
// the importance sampling waits for them (hl_render_rays_u_event).  What is left of the gap (kernel trace, `scripts/rocpd_timeline.py`): an evaluate launch
// that overlaps the walk takes 14.4 instead of 12 ms - its 1 024 workgroups do not share the walk's CU, so four of them run a fifth round - whichever of the
// two evaluate launches that is (holding the walk back until the caller's queued work has finished moves the cost from the fine to the coarse pass, 31.2 against
// 30.6 ms; a raised wave priority changes nothing).
#include "hl_common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr unsigned MT_A = 0x9908b0dfu, MT_UP = 0x80000000u, MT_LO = 0x7fffffffu;

__device__ __forceinline__ unsigned mt_twist(unsigned u, unsigned v) { return (((u & MT_UP) | (v & MT_LO)) >> 1) ^ ((v & 1u) ? MT_A : 0u); }

__device__ __forceinline__ unsigned &a0r(const unsigned &v) { return const_cast<unsigned &>(v); }
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// st_in: 624 state words; pos: words of the current block already handed out (624: the block is used up - also the freshly seeded generator);
// raw[0..n): the next n words of the stream, UNTEMPERED (k_mt19937_finish turns them into uniforms in place); st_out: 624 words + the new position.
// n * 4 must fit a buffer descriptor's 32-bit range (the host splits longer requests).  st_out may be st_in.
__global__ __launch_bounds__(256) void k_mt19937_walk(const unsigned *__restrict__ st_in, int pos, unsigned *__restrict__ raw, long n,
                                                      unsigned *__restrict__ st_out) {
    __shared__ unsigned buf[2][MT_N];
    const int tid = threadIdx.x;
    for (int i = tid; i < MT_N; i += 256) buf[0][i] = st_in[i];
    __syncthreads();
    int cur = 0;
    long done = 0;
    if (pos < MT_N) {   // the rest of the block the host generator was in
        const long cnt = min(n, (long)(MT_N - pos));
        for (int i = tid; i < cnt; i += 256) raw[i] = buf[0][pos + i];
        done = cnt;
        pos += (int)cnt;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)raw, (short)0, (int)(n * 4), 0x00020000);
    // One regeneration per barrier: thread t < 227 forms words t, t + 227 and t + 454 - the lag-397 operand of the second and third is the word
    // the SAME thread has just formed (x[k + 397 - 624] = x[k - 227]), so only values of the previous block are read from LDS (all reads of a
    // thread are issued together: one LDS latency per block); word 623 twists with the NEW word 0, which its thread forms again for itself.
    const int k2 = tid + (MT_N - MT_M), k3 = tid + 2 * (MT_N - MT_M);
    const bool act = tid < MT_N - MT_M, has3 = k3 < MT_N;
    const int k3c = has3 ? k3 : MT_N - 1, k3c1 = k3 + 1 < MT_N ? k3 + 1 : MT_N - 1;
    while (done < n) {
        const unsigned *o = buf[cur];
        unsigned *w = buf[cur ^ 1];
        const long left = n - done;
        if (act) {
            // (every read unconditional, indices clamped: ONE batch of LDS requests per block - a read behind a branch costs a second round trip)
            const unsigned a0 = o[tid], a1 = o[tid + 1], am = o[tid + MT_M], b0 = o[k2], b1 = o[k2 + 1];
            const unsigned c0 = o[k3c], c1o = o[k3c1];
            const unsigned w0 = o[0], w1 = o[1], wm = o[MT_M];
            asm volatile("" : "+v"(a0r(a0)), "+v"(a0r(c0)), "+v"(a0r(c1o)), "+v"(a0r(w0)), "+v"(a0r(w1)), "+v"(a0r(wm)));
            const unsigned n0 = wm ^ mt_twist(w0, w1);                   // the new word 0 (broadcast reads)
            const unsigned y1 = am ^ mt_twist(a0, a1);
            const unsigned y2 = y1 ^ mt_twist(b0, b1);
            w[tid] = y1;
            w[k2] = y2;
            const int so = (int)(done * 4);                               // (wave-uniform: a scalar offset; words behind raw[n - 1] fall to the bounds check)
            __builtin_amdgcn_raw_buffer_store_b32(y1, rs, tid * 4, so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(y2, rs, k2 * 4, so, 0);
            if (has3) {
                const unsigned y3 = y2 ^ mt_twist(c0, k3 == MT_N - 1 ? n0 : c1o);
                w[k3] = y3;
                __builtin_amdgcn_raw_buffer_store_b32(y3, rs, k3 * 4, so, 0);
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
    __syncthreads();
    for (int i = tid; i < MT_N; i += 256) st_out[i] = buf[cur][i];
    if (tid == 0) st_out[MT_N] = (unsigned)pos;
}

// raw state words -> uniforms, in place (the same four bytes): u = (tempered word & (2^24 - 1)) * 2^-24
__global__ __launch_bounds__(256) void k_mt19937_finish(float *__restrict__ out, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const unsigned y = mt_temper(__builtin_bit_cast(unsigned, out[i]));
        out[i] = (float)(y & 0xffffffu) * 5.9604644775390625e-08f;   // (24 bits: exact in fp32) * 2^-24
    }
}

}  // namespace

static int64_t g_mt_piece = (int64_t)1 << 28;
extern "C" int hl_debug_set_mt19937_piece(int64_t words) { g_mt_piece = words > 0 ? words : (int64_t)1 << 28; return HL_OK; }

extern "C" int hl_mt19937_uniform(const uint32_t *state, int pos, float *out, int64_t n, uint32_t *state_out, void *stream) {
    HL_REQUIRE(state && out && state_out && n >= 0 && pos >= 0 && pos <= MT_N, "hl_mt19937_uniform: bad argument");
    // the walk addresses its output through a buffer descriptor (32-bit byte range): requests beyond 2^28 words continue from the state the piece before left
    const int64_t piece = g_mt_piece;
    const uint32_t *st_in = state;
    int64_t done = 0;
    do {
        const int64_t cnt = n - done < piece ? n - done : piece;
        hipLaunchKernelGGL(k_mt19937_walk, dim3(1), dim3(256), 0, (hipStream_t)stream, st_in, pos, reinterpret_cast<unsigned *>(out) + done, (long)cnt, state_out);
        int rc = hl::check_launch("k_mt19937_walk");
        if (rc) return rc;
        // the position behind cnt words (what the kernel stores in state_out[624]): the rest of the current block first, whole blocks after that
        if (cnt > 0) {
            const int64_t head = MT_N - pos;
            pos = cnt <= head ? pos + (int)cnt : (int)((cnt - head - 1) % MT_N) + 1;
        }
        st_in = state_out;
        done += cnt;
    } while (done < n);
    if (n > 0) {
        long g = (long)((n + 255) / 256);
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(k_mt19937_finish, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, out, (long)n);
        return hl::check_launch("k_mt19937_finish");
    }
    return HL_OK;
}
