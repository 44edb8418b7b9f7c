// Tri-plane NeRF volume renderer for MI355X (gfx950): hand-written HIP, fp32 MFMA.
//
// Replaces, for use_canonical_space=False / test mode, the reference's
//   Renderer.render / render_core / up_sample / NeRF_network   human_diffusion/NeRF/renderer.py:134-281
//   sample_from_planes / project_onto_planes / sample_pdf       human_diffusion/NeRF/renderer.py:486-563
//   PositionalEncoding.forward                                  human_diffusion/NeRF/fields.py:45-85
// (the reference launches ~60 stock PyTorch kernels per chunk and materialises every intermediate).
//
// Three launches per ray batch, everything else stays on chip:
//   k_march<false>  coarse pass: tri-plane gather -> density MLP            -> sigma (R,N)
//   k_importance    per ray: weights -> inverse CDF with the caller's u -> sort -> z_all (R,2N)
//   k_march<true>   fine pass:   tri-plane gather -> full MLP -> alpha compositing -> rgb/acc/depth
//
// MLP on the matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32, an fmaf chain):
//   The layer is evaluated TRANSPOSED: D[unit][ray] += W[unit][k] * H[k][ray].  With the 32x32x2
//   fragment layout the accumulator of lane l holds, for ray (l & 31), the units
//   (r&3) + 8*(r>>2) + 4*(l>>5) — which is exactly the shape of a B operand (lane l supplies
//   B[k = l>>5][ray = l&31]) if step r pairs unit u_lo(r) on lanes 0-31 with u_lo(r)+4 on lanes
//   32-63.  The weights (A operand) are pre-packed in that k order, so activations never leave
//   the register file between layers: no LDS round trip, no transposes, softplus in place.
//   One wave owns 32 rays (two lanes per ray: the halves split the 27 features / 27 view
//   encodings and the hidden units); a workgroup is 8 waves = 256 rays and streams the packed
//   weights (17 x 16 KB per sample) from L2 through a two-slot LDS ring, one barrier per chunk.
//
// Further down in this file: evaluate-once schedule (k_march<.., STORE> + k_composite), canonical-space deformation
// (k_deform_rays_cull / k_deform_rays / k_deform_points, renderer.py:52-132), per-view ray generation (k_camera_rays), and the
// training backward (k_march<.., ACTS>, k_composite_wave, k_mlp_bwd, k_plane_scatter, k_wgrad; recon_NeRF/run_nerf_batch.py:236-265).
#include "hl_common.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

// ---------------------------------------------------------------------------------------------
// packed layout
// ---------------------------------------------------------------------------------------------
constexpr int CHUNK_FLOATS = 4096;  // 16 KB: [tile t][step/4][lane 64][4 steps]
constexpr int NCH_FULL = 17;
constexpr int NCH_COARSE = 10;
constexpr int SMALL_FLOATS = 1024;
constexpr int PACKED_FLOATS = NCH_FULL * CHUNK_FLOATS + SMALL_FLOATS;
// offsets inside the small-parameter block (floats); [t][half][16] lane-order tables
constexpr int SM_B0 = 0, SM_B1 = 128, SM_B2 = 256, SM_BF = 384, SM_BV = 512, SM_AW = 576, SM_RW = 704,
              SM_AB = 896, SM_RB = 897;
// [SM_SC + l], l = 0..4 (pts_linears.0 / .1 / .2, feature_linear, views_linear): the power of two the fp16x2 weight planes of layer l are multiplied by
// (k_mlp_scales_h2, round 6); [SM_SC + 8 + l] its inverse
constexpr int SM_SC = 904;

struct ChunkDesc {
    int w;      // 0 pts0, 1 pts1, 2 pts2, 3 feat, 4 views
    int ld;     // row length of that weight
    int col0;   // first input column of this part
    int kind;   // 0 feature part, 1 hidden part, 2 view-encoding part
    int base;   // hidden part: first step (k-pair) covered
    int nt;     // 32-unit output tiles
    int nsteps; // padded steps in the chunk image
};
__constant__ ChunkDesc c_chunks[NCH_FULL] = {
    {0, 27, 0, 0, 0, 4, 16},                                                                      // L0
    {1, 128, 0, 1, 0, 4, 16},  {1, 128, 0, 1, 16, 4, 16}, {1, 128, 0, 1, 32, 4, 16}, {1, 128, 0, 1, 48, 4, 16},  // L1
    {2, 155, 0, 0, 0, 4, 16},                                                                     // L2 (features)
    {2, 155, 27, 1, 0, 4, 16}, {2, 155, 27, 1, 16, 4, 16}, {2, 155, 27, 1, 32, 4, 16}, {2, 155, 27, 1, 48, 4, 16},
    {3, 128, 0, 1, 0, 4, 16},  {3, 128, 0, 1, 16, 4, 16}, {3, 128, 0, 1, 32, 4, 16}, {3, 128, 0, 1, 48, 4, 16},  // feature
    {4, 155, 0, 1, 0, 2, 32},  {4, 155, 0, 1, 32, 2, 32},                                         // views (feature)
    {4, 155, 128, 2, 0, 2, 16},                                                                   // views (dir enc)
};

// unit index carried by accumulator register r (0..15) of tile t on lane-half h
__host__ __device__ inline int unit_of(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

struct PackArgs {
    const float *w[5];
    const float *b[5];  // pts0,pts1,pts2,feat,views
    const float *alpha_w, *alpha_b, *rgb_w, *rgb_b;
    float *out;
};

__global__ void k_pack_mlp(PackArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= PACKED_FLOATS) return;
    float v = 0.f;
    if (idx < NCH_FULL * CHUNK_FLOATS) {
        const int c = idx / CHUNK_FLOATS, e = idx % CHUNK_FLOATS;
        const ChunkDesc d = c_chunks[c];
        const int ns4 = d.nsteps / 4;
        const int k4 = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
        const int s4 = rest % ns4, t = rest / ns4;
        if (t < d.nt) {
            const int step = s4 * 4 + k4, half = lane >> 5, out = 32 * t + (lane & 31);
            int in = -1;
            if (d.kind == 0) {
                const int k = step + 15 * half;
                if (step < 15 && k < 27) in = k;
            } else if (d.kind == 1) {
                const int sp = d.base + step;
                in = unit_of(sp >> 4, sp & 15, half);
            } else {
                const int k = step + 14 * half;
                if (step < 14 && k < 27) in = k;
            }
            if (in >= 0) v = a.w[d.w][out * d.ld + d.col0 + in];
        }
    } else {
        const int s = idx - NCH_FULL * CHUNK_FLOATS;
        if (s < SM_AW) {  // biases
            const int layer = s < SM_BV ? s / 128 : 4;
            const int o = s - (layer < 4 ? layer * 128 : SM_BV);
            const int r = o & 15, h = (o >> 4) & 1, t = o >> 5;
            v = a.b[layer][unit_of(t, r, h)];
        } else if (s < SM_RW) {
            const int o = s - SM_AW, r = o & 15, h = (o >> 4) & 1, t = o >> 5;
            v = a.alpha_w[unit_of(t, r, h)];
        } else if (s < SM_AB) {
            const int o = s - SM_RW, r = o & 15, h = (o >> 4) & 1, t = (o >> 5) & 1, c = o >> 6;
            v = a.rgb_w[c * 64 + unit_of(t, r, h)];
        } else if (s == SM_AB) {
            v = a.alpha_b[0];
        } else if (s < SM_RB + 3) {
            v = a.rgb_b[s - SM_RB];
        } else if (s >= SM_SC && s < SM_SC + 16) {
            v = 1.f;                                    // (k_mlp_scales_h2 overwrites the entries it owns)
        }
    }
    a.out[idx] = v;
}

// Scales of the fp16x2 weight planes (round 6).  The low plane w - fp16(w) of a weight below 2^-3 is subnormal in fp16 (absolute precision 2^-25): at nn.Linear's
// default initialisation (|w| <= 0.09 ... 0.19) the pair already keeps 2^-21 instead of 2^-22, and an MLP whose weights are 2^-8 of that would be down to TF32's
// precision.  So every LAYER's weights are multiplied, before the split, by the power of two that puts the layer's largest |w| into [2^12, 2^13) (x log2(e) <
// 2^14 in front of a softplus), the bias that initialises the accumulators likewise, and the kernel multiplies the accumulators by the inverse on their way into
// the softplus (or into the next operand): exact, one v_mul per accumulator register.  One scale per layer, not per unit: the accumulators of a lane hold 16
// different units.  grid = 5 layers.
__global__ void k_mlp_scales_h2(PackArgs a, float *dst) {   // dst[l] = the scale of layer l, dst[8 + l] its inverse
    __shared__ float red[256];
    const int l = blockIdx.x;
    const int n = l == 0 ? 128 * 27 : (l == 1 ? 128 * 128 : (l == 2 ? 128 * 155 : (l == 3 ? 128 * 128 : 64 * 155)));
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(a.w[l][i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        m = red[0];
        int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xff) - 127;     // floor(log2 m) for a normal m
        float sc = 1.f, inv = 1.f;
        if (m > 0.f && e < 128) {
            e = e < -100 ? -100 : e;
            sc = __builtin_bit_cast(float, (unsigned)(12 - e + 127) << 23);
            inv = __builtin_bit_cast(float, (unsigned)(e - 12 + 127) << 23);
        }
        dst[l] = sc;
        dst[8 + l] = inv;
    }
}

// planes (3,9,H,W) -> [q = plane*3 + group][y][x][4] (3 channels + 0)
__global__ void k_pack_planes(const float *__restrict__ src, float4 *__restrict__ dst, int H, int W) {
    const int64_t n = (int64_t)9 * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = i / ((int64_t)H * W), yx = i % ((int64_t)H * W);
        const float *s = src + (q * 3) * (int64_t)H * W + yx;
        dst[i] = make_float4(s[0], s[(int64_t)H * W], s[2 * (int64_t)H * W], 0.f);
    }
}

// ---------------------------------------------------------------------------------------------
// math helpers (file is compiled with -ffp-contract=off: mul/add sequences stay as the
// reference's eager PyTorch ops evaluate them)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_hidden(float x) {
    // F.softplus(beta=1, threshold=20) = max(x,0) + ln(1 + exp(-|x|)).  Raw v_exp_f32 / v_log_f32 (2^x, log2 x,
    // ~1 ulp): the argument of exp2 is <= 0 and the argument of log2 lies in [1,2], so none of the denormal /
    // range fix-ups of the libm-style wrappers (13 extra VALU ops per call) are needed.  abs error ~1e-7.
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(x));
    return fmaf(0.693147180559945309f, __builtin_amdgcn_logf(1.f + e), fmaxf(x, 0.f));
}
__device__ __forceinline__ float softplus_exact(float x) {
    // density softplus feeds 1-exp(-sp*1e10) on the last sample: keep full relative accuracy for x << 0
    return x > 20.f ? x : log1pf(expf(x));
}
// Compositing arithmetic of the inference paths - k_march's own compositing, k_composite and the one-pass fine launch share it, so their images stay bit-identical
// (round 6).  Inside a ray: the raw v_exp_f32 / v_log_f32 / v_rcp_f32 forms (~1 ulp each; alpha to ~1e-9 absolute): the one-pass launch composites inside the MLP
// kernel, where the libm forms - ~200 instructions per sample, times the divergence of the per-lane merge loop - cost 2 ms per 512x512 view.  The LAST sample of a ray
// (distance 1e10, renderer.py:213) keeps the exact forms: alpha = 1 - exp(-softplus(x) 1e10) needs softplus to full RELATIVE accuracy for x << 0.
__device__ __forceinline__ float comp_alpha(float sraw, float dist) { return 1.f - __builtin_amdgcn_exp2f(-1.44269504088896341f * (softplus_hidden(sraw) * dist)); }
__device__ __forceinline__ float comp_alpha_last(float sraw) { return 1.f - expf(-softplus_exact(sraw) * 1e10f); }
__device__ __forceinline__ float comp_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v)); }
__device__ __forceinline__ f32x16 softplus16(f32x16 v) {
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = softplus_hidden(v[i]);
    return o;
}

// 16 k-steps (4 groups of 4) of NT output tiles.  A fragments: one ds_read_b128 = 4 steps of one tile.
template <int NT, int NS4, int NREAL>
__device__ __forceinline__ void mma16(f32x16 (&acc)[NT], const f32x16 b, const f32x4 *__restrict__ ldsA, int s4base,
                                      int lane) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        if (s4 * 4 >= NREAL) break;
        f32x4 a[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = ldsA[(t * NS4 + s4base + s4) * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (s4 * 4 + k < NREAL) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][k], b[s4 * 4 + k], acc[t], 0, 0, 0);
            }
        }
    }
}

// accumulator init = bias in lane order ([t][half][16] table in LDS)
template <int NT>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[NT], const float *__restrict__ tbl, int half) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 *p = reinterpret_cast<const f32x4 *>(tbl + (t * 2 + half) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = p[q];
            acc[t][4 * q + 0] = v[0];
            acc[t][4 * q + 1] = v[1];
            acc[t][4 * q + 2] = v[2];
            acc[t][4 * q + 3] = v[3];
        }
    }
}

template <int NT>
__device__ __forceinline__ float dot_lane(const f32x16 (&h)[NT], const float *__restrict__ tbl, int half) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 *p = reinterpret_cast<const f32x4 *>(tbl + (t * 2 + half) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = p[q];
            s = fmaf(h[t][4 * q + 0], v[0], s);
            s = fmaf(h[t][4 * q + 1], v[1], s);
            s = fmaf(h[t][4 * q + 2], v[2], s);
            s = fmaf(h[t][4 * q + 3], v[3], s);
        }
    }
    return s + __shfl_xor(s, 32);  // the two halves of a ray hold disjoint units
}

struct MarchArgs {
    const float *packed;   // hl_render_mlp_pack output
    const float4 *planes;  // hl_planes_pack output
    int H, W;
    const float *bounds;  // (2,3) device
    const float *rays_o, *rays_d, *near, *far;
    const float *z;  // depths: null (linspace), caller rows (R,S), or the tile-major workspace [R/32][S][32]
    int z_tiled;
    long long R;
    int S;  // samples marched per ray
    unsigned flags;
    float *sigma_out;            // coarse
    float4 *vals_out;            // STORE: raw (sigma, r, g, b) per sample, tile-major [R/32][S][32]
    const float4 *pts_c, *dirs_c;  // POINTS: canonical-space sample points / view directions, tile-major [R/32][S][32] (xyz, w unused)
    float *rgb, *acc, *depth;    // fine
    // training: ACTS stores every activation of the MLP, transposed - row = unit (ROW_* below), column =
    // sample point in the pass's tile-major order - for the backward kernels; k_mlp_bwd's own inputs / outputs follow
    float *act;
    long long act_stride, act_off;   // floats per row; first column of this pass
    const float *bwd_packed;         // hl_render_mlp_pack_bwd output
    const float4 *d_rec;             // dL/d(sigma_raw, r_raw, g_raw, b_raw) per sample, same layout as vals_out
    float *del;                      // row = unit (DROW_*), column = sample point: layer deltas for the weight gradients
    long long del_stride, del_off;
    float *dplanes;                  // (27, H, W) gradient of the tri-plane, accumulated atomically
    int s_per;                       // ACTS / k_mlp_bwd: samples per workgroup; blockIdx.y selects the range (a fitting batch has few
                                     // rays - 2048 = 8 workgroups of 256 - so the launch is spread over the samples as well)
    // one-pass fine launch (k_march_plw<., false, true>, round 6): the wave that owns a 32-ray tile draws its importance depths from the coarse records, evaluates
    // them and composites coarse + new samples in depth order as it goes - no fine records, no merge kernel
    const float4 *fz_vc;             // the coarse launch's raw records, tile-major [R/32][fz_N][32]
    const float *fz_u;               // sample_pdf's uniforms, rows (R, S)
    float *fz_zn;                    // scratch for the new depths, sorted, tile-major [R/32][S][32]: written and read back by the tile's own wave
    int fz_N;                        // coarse samples per ray (<= 128; their depths are linspace(near, far))
};
// rows of the activation matrix ([features | hidden1] and [feature_linear | view encoding] are the concatenations the network feeds
// to pts_linears.2 and views_linear, so their weight gradients are single products)
constexpr int ROW_F = 0, ROW_X1 = 27, ROW_X0 = 155, ROW_X2 = 283, ROW_Y = 411, ROW_EV = 539, ROW_V = 566, ACT_ROWS = 630;
constexpr int DROW_X0 = 0, DROW_X1 = 128, DROW_X2 = 256, DROW_Y = 384, DROW_V = 512, DROW_REC = 576, DROW_DF = 580, DROW_DFT = 607, DEL_ROWS = 634;

// Hidden (inline-asm) global store / atomic add: see the note at the record store of k_march - a compiler-visible VMEM write in the
// sample loop turns every counted vmcnt wait of the weight ring into vmcnt(0).
// The matrices are addressed through a buffer descriptor: per-lane byte offset (column, and the lane-half's share of the row) in a
// VGPR, the row's byte offset in an SGPR - 64-bit per-row addresses would be hoisted out of the sample loop by the hundred and
// spilled.  Offsets are 32-bit: a matrix is kept below 4 GiB (checked by the launchers).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 matrix_rsrc(const void *p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void hidden_store(i32x4 rs, unsigned voff, unsigned soff, float v) {
    // The row offset is added on the VALU and the scalar offset left at 0: the hazard recogniser does not look into inline asm, and an
    // SGPR written by the SALU (or by a v_readlane of a spilled descriptor) needs 5 wait states before a VMEM instruction reads it -
    // hence also the s_nop (317 stores per sample and lane: 2 % of the sample's MFMA time).
    const unsigned o = voff + soff;
    asm volatile("s_nop 4\n\tbuffer_store_dword %0, %1, %2, 0 offen" : : "v"(v), "v"(o), "s"(rs) : "memory");
}
__device__ __forceinline__ void hidden_atomic_add(float *p, float v) { asm volatile("global_atomic_add_f32 %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
// rows row0 + unit_of(t, r, half): voff must already hold the column and half * 4 rows
template <int NT>
__device__ __forceinline__ void store_rows(i32x4 rs, unsigned voff, unsigned stride4, int row0, const f32x16 (&h)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) hidden_store(rs, voff, (unsigned)(row0 + unit_of(t, r, 0)) * stride4, h[t][r]);
}

// torch.linspace(0,1,N)[s] (CPU/CUDA kernels are symmetric about the midpoint)
__device__ __forceinline__ float linspace01(int s, int N) {
    const float step = 1.0f / (float)(N - 1);
    return s < N / 2 ? step * (float)s : 1.0f - step * (float)(N - 1 - s);
}

// STORE (with FULL): evaluate the full MLP at every sample and write the raw outputs instead of compositing - the two halves
// of the sample set (coarse, importance) are evaluated once each and merged by k_composite.
// NWV waves (of 32 rays) per workgroup share one weight ring.  8: one workgroup per CU (2 waves/SIMD, all in lock step through the
// 19 barriers of a sample); 4: two independent workgroups per CU whose activation / gather phases overlap each other's MFMAs.
// POINTS (with STORE): sample positions and view directions are read per sample (canonical-space rendering: k_deform_rays wrote
// them) instead of being o + d*z and the ray's direction; the view-direction encoding is then evaluated per sample.
template <bool FULL, bool STORE = false, int NWV = 8, bool POINTS = false, bool ACTS = false>
__global__ __launch_bounds__(NWV * 64, 2) void k_march(const MarchArgs a) {
    static_assert(!POINTS || (FULL && STORE), "POINTS is an evaluate-pass mode");
    static_assert(!ACTS || (FULL && STORE), "ACTS is an evaluate-pass mode");
    constexpr int NT = NWV * 64, NST = 1024 / NT;   // threads; float4 per thread and 16 KB chunk
    __shared__ __attribute__((aligned(16))) float lds[2 * CHUNK_FLOATS + SMALL_FLOATS];
    constexpr int NCH = FULL ? NCH_FULL : NCH_COARSE;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    // Workgroup b runs on XCD b % 8 (observed; speed only): hand every XCD a contiguous run of 256-ray groups so
    // its L2 sees one band of the image (and of the tri-planes) instead of rows from everywhere.
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const long long wg = (long long)xcd * q8 + (xcd < r8 ? xcd : r8) + (blockIdx.x >> 3);
    const long long tile = wg * NWV + (tid >> 6);           // 32-ray tile of this wave
    const long long ray = tile * 32 + (lane & 31);
    const bool valid = ray < a.R;
    const long long rc = valid ? ray : a.R - 1;
    // workspace arrays are tile-major [tile][sample][32 rays]: one 128-byte line per wave and sample
    // (waves past the last tile of a ragged batch clamp to it for reads and never store)
    const long long tiles_n = (a.R + 31) / 32;
    const long long zt_base = (tile < tiles_n ? tile : tiles_n - 1) * 32 * (long long)a.S + (lane & 31);

    f32x4 *ldsv = reinterpret_cast<f32x4 *>(lds);
    const float *small = lds + 2 * CHUNK_FLOATS;
    const f32x4 *gw = reinterpret_cast<const f32x4 *>(a.packed);
    for (int i = tid; i < SMALL_FLOATS / 4; i += NT) ldsv[2 * CHUNK_FLOATS / 4 + i] = gw[NCH_FULL * CHUNK_FLOATS / 4 + i];
    // weight ring: chunk 0 -> slot 0, chunk 1 -> slot 1, chunk 2 staged in registers
#pragma unroll
    for (int q = 0; q < 2 * NST; ++q) ldsv[q * NT + tid] = gw[q * NT + tid];
    // the chunk prefetches go through a buffer descriptor: per-lane offset tid*16 + a compile-time scalar chunk offset, instead of
    // 17 different 64-bit per-lane addresses (which the allocator spilled and reloaded from scratch inside the sample loop)
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)a.packed, (short)0, PACKED_FLOATS * 4, 0x00020000);
    const int wv = tid * 16;
    auto ldw = [&](int f4_index) -> f32x4 {   // float4 index of lane 0; this lane reads index + tid
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wv, f4_index * 16, 0));
    };
    f32x4 st[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) st[q] = ldw(2048 + q * NT);
    int cur = 0;  // float4 offset of the slot holding the chunk being consumed

    const float ox = a.rays_o[rc * 3 + 0], oy = a.rays_o[rc * 3 + 1], oz = a.rays_o[rc * 3 + 2];
    const float dx = a.rays_d[rc * 3 + 0], dy = a.rays_d[rc * 3 + 1], dz = a.rays_d[rc * 3 + 2];
    const float nr = a.near[rc], fr = a.far[rc];
    const int S = a.S;
    const float offH = (float)(1.0 / (double)a.H);
    const float bmin0 = a.bounds[0], bmin1 = a.bounds[1], bmin2 = a.bounds[2];
    const float bext0 = a.bounds[3] - bmin0, bext1 = a.bounds[4] - bmin1, bext2 = a.bounds[5] - bmin2;

    // view-direction encoding, this half's 14 of the 27 (+1 pad) entries   [fields.py:54-85]
    // entry k: k<3 raw component; else j=(k-3)/3, comp=(k-3)%3, sin(vd*2^(j/2) + (j&1)*pi/2).  Lanes 0-31 hold k = s, lanes 32-63
    // k = s + 14: the argument is selected per half BEFORE the sine, so each lane evaluates 14 sines, not 28.
    auto view_encode = [&](float v0, float v1, float v2) -> f32x16 {
        const float vd[3] = {v0, v1, v2};
        f32x16 e;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float val = 0.f;
            if (s < 14) {
                const int kl = s, kh = s + 14;                                   // compile-time per unrolled s
                const int jl = (kl - 3) / 3, cl = (kl - 3) % 3, jh = (kh - 3) / 3, ch = (kh - 3) % 3;
                const float argl = kl < 3 ? 0.f : ((jl & 1) ? 1.57079632679489661923f : 0.f) + vd[kl < 3 ? 0 : cl] * (float)(1 << (jl >> 1));
                const float argh = ((jh & 1) ? 1.57079632679489661923f : 0.f) + vd[ch] * (float)(1 << (jh >> 1));
                if (kl < 3) {                    // low half: raw component; high half: a sine
                    const float sh = sinf(argh);
                    val = half ? sh : vd[kl];
                } else if (kh >= 27) {           // high half: padding
                    const float sl = sinf(argl);
                    val = half ? 0.f : sl;
                } else {
                    val = sinf(half ? argh : argl);
                }
            }
            e[s] = val;
        }
        return e;
    };
    f32x16 ev;
    if constexpr (FULL && !POINTS) {
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        ev = view_encode(dx / nrm, dy / nrm, dz / nrm);
    }

    const unsigned act_stride4 = ACTS ? (unsigned)a.act_stride * 4u : 0u;
    const i32x4 act_rs = matrix_rsrc(a.act, ACTS ? (unsigned)ACT_ROWS * act_stride4 : 0u);
    float T = 1.f, acc_w = 0.f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f;
    float zc;  // depth of the current sample
    int s_lo = 0, s_hi = S;
    if constexpr (ACTS) {
        s_lo = (int)blockIdx.y * a.s_per;
        s_hi = min(S, s_lo + a.s_per);
    }
    if (a.z) zc = a.z_tiled ? a.z[zt_base + 32LL * s_lo] : a.z[rc * S + s_lo];
    else zc = nr * (1.f - linspace01(s_lo, S)) + fr * linspace01(s_lo, S);

    __syncthreads();

#define HL_CHUNK_ADVANCE(cnext2)                                              \
    __syncthreads();                                                          \
    cur ^= 1024;                                                              \
    _Pragma("unroll") for (int q_ = 0; q_ < NST; ++q_) ldsv[(cur ^ 1024) + q_ * NT + tid] = st[q_];          \
    _Pragma("unroll") for (int q_ = 0; q_ < NST; ++q_) st[q_] = ldw((cnext2) * 1024 + q_ * NT);

    // Ring invariant while chunk g is consumed: slot `cur` holds g, the other slot holds (or is
    // being filled with) g+1, the staging registers hold (or are receiving) g+2.
    // HL_CHUNK_ADVANCE(g+2), executed when moving on to chunk g: (1) barrier - every wave is done
    // with g-1 and the writes of g are visible, (2) flip `cur`, (3) write the staged chunk g+1 into
    // the slot g-1 just released, (4) start loading g+2 from L2 (lands during g's MFMAs).
    for (int s = s_lo; s < s_hi; ++s) {
        // ---- next depth (needed for the section length) ----
        float zn = 0.f;
        if (s + 1 < S) {
            if (a.z) zn = a.z_tiled ? a.z[zt_base + 32LL * (s + 1)] : a.z[rc * S + s + 1];
            else { const float t = linspace01(s + 1, S); zn = nr * (1.f - t) + fr * t; }
        }
        // ---- tri-plane features of this half  [renderer.py:502-531] ----
        float px, py, pz;
        if constexpr (POINTS) {
            const float4 pc = a.pts_c[zt_base + 32LL * s], dc = a.dirs_c[zt_base + 32LL * s];
            px = pc.x; py = pc.y; pz = pc.z;
            ev = view_encode(dc.x, dc.y, dc.z);    // canonical directions are used as they are (renderer.py:150)
        } else {
            px = ox + dx * zc; py = oy + dy * zc; pz = oz + dz * zc;
        }
        const float nx = 2.f * (px - bmin0) / bext0 - 1.f;
        const float ny = 2.f * (py - bmin1) / bext1 - 1.f;
        const float nz = 2.f * (pz - bmin2) / bext2 - 1.f;
        f32x16 f;
        f[15] = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int qlo = i, qhi = (i + 5 > 8) ? 8 : i + 5;
            const int q = half ? qhi : qlo;
            const int p = half ? qhi / 3 : qlo / 3, g = half ? qhi % 3 : qlo % 3;
            float gu = (p == 2) ? nz : nx;
            float gv = (p == 1) ? nz : ny;
            gu = (g == 1) ? gu + offH : gu;
            gv = (g == 2) ? gv + offH : gv;
            const float ix = ((gu + 1.f) * (float)a.W - 1.f) / 2.f;
            const float iy = ((gv + 1.f) * (float)a.H - 1.f) / 2.f;
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float x1f = x0f + 1.f, y1f = y0f + 1.f;
            float w_nw = (x1f - ix) * (y1f - iy), w_ne = (ix - x0f) * (y1f - iy);
            float w_sw = (x1f - ix) * (iy - y0f), w_se = (ix - x0f) * (iy - y0f);
            const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
            const bool vx0 = (x0 >= 0) & (x0 < a.W), vx1 = (x1 >= 0) & (x1 < a.W);
            const bool vy0 = (y0 >= 0) & (y0 < a.H), vy1 = (y1 >= 0) & (y1 < a.H);
            w_nw = (vx0 & vy0) ? w_nw : 0.f;
            w_ne = (vx1 & vy0) ? w_ne : 0.f;
            w_sw = (vx0 & vy1) ? w_sw : 0.f;
            w_se = (vx1 & vy1) ? w_se : 0.f;
            const int cx0 = min(max(x0, 0), a.W - 1), cx1 = min(max(x1, 0), a.W - 1);
            const int cy0 = min(max(y0, 0), a.H - 1), cy1 = min(max(y1, 0), a.H - 1);
            const float4 *pl = a.planes + (long long)q * a.H * a.W;
            const float4 t_nw = pl[cy0 * a.W + cx0], t_ne = pl[cy0 * a.W + cx1];
            const float4 t_sw = pl[cy1 * a.W + cx0], t_se = pl[cy1 * a.W + cx1];
            const bool live = half ? (i + 5 <= 8) : true;
            const float r0 = t_nw.x * w_nw + t_ne.x * w_ne + t_sw.x * w_sw + t_se.x * w_se;
            const float r1 = t_nw.y * w_nw + t_ne.y * w_ne + t_sw.y * w_sw + t_se.y * w_se;
            const float r2 = t_nw.z * w_nw + t_ne.z * w_ne + t_sw.z * w_sw + t_se.z * w_se;
            f[3 * i + 0] = live ? r0 : 0.f;
            f[3 * i + 1] = live ? r1 : 0.f;
            f[3 * i + 2] = live ? r2 : 0.f;
        }

        const bool act_on = ACTS && tile * 32 < a.R;
        unsigned acol = 0;       // byte offset of this sample point's column (+ this half's 4 rows) in the activation matrix
        if constexpr (ACTS) {
            const unsigned col4 = (unsigned)(a.act_off + zt_base + 32LL * s) * 4u;
            acol = col4 + (unsigned)half * 4u * act_stride4;
            if (act_on) {
#pragma unroll
                for (int j = 0; j < 15; ++j)
                    if (j + 15 * half < 27) hidden_store(act_rs, col4 + (unsigned)half * 15u * act_stride4, (unsigned)(ROW_F + j) * act_stride4, f[j]);
#pragma unroll
                for (int j = 0; j < 14; ++j)
                    if (j + 14 * half < 27) hidden_store(act_rs, col4 + (unsigned)half * 14u * act_stride4, (unsigned)(ROW_EV + j) * act_stride4, ev[j]);
            }
        }
        // ---- MLP  [renderer.py:134-156] ----
        f32x16 X[4], Y[4];
        // L0: features -> X (chunk 0)
        load_bias<4>(X, small + SM_B0, half);
        mma16<4, 4, 15>(X, f, ldsv + cur, 0, lane);
        // L1: softplus(X) -> Y (chunks 1-4)
        load_bias<4>(Y, small + SM_B1, half);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            HL_CHUNK_ADVANCE((k + 3) % NCH)
            X[k] = softplus16(X[k]);
            mma16<4, 4, 16>(Y, X[k], ldsv + cur, 0, lane);
        }
        if constexpr (ACTS) { if (act_on) store_rows<4>(act_rs, acol, act_stride4, ROW_X0, X); }
        // L2: [features, softplus(Y)] -> X (chunks 5, 6-9)
        HL_CHUNK_ADVANCE(7 % NCH)
        load_bias<4>(X, small + SM_B2, half);
        mma16<4, 4, 15>(X, f, ldsv + cur, 0, lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            HL_CHUNK_ADVANCE((k + 8) % NCH)
            Y[k] = softplus16(Y[k]);
            mma16<4, 4, 16>(X, Y[k], ldsv + cur, 0, lane);
        }
        if constexpr (ACTS) { if (act_on) store_rows<4>(act_rs, acol, act_stride4, ROW_X1, Y); }
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = softplus16(X[k]);
        if constexpr (ACTS) { if (act_on) store_rows<4>(act_rs, acol, act_stride4, ROW_X2, X); }
        const float sigma_raw = dot_lane<4>(X, small + SM_AW, half) + small[SM_AB];

        if constexpr (!FULL) {
            if (half == 0 && tile * 32 < a.R) a.sigma_out[zt_base + 32LL * s] = sigma_raw;   // tiles are padded to 32 rays
            HL_CHUNK_ADVANCE(2)  // back to chunk 0 for the next sample (chunk 1 staged, stage chunk 2)
        } else {
            // feature_linear (no activation): X -> Y (chunks 10-13)
            load_bias<4>(Y, small + SM_BF, half);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                HL_CHUNK_ADVANCE((k + 12) % NCH)
                mma16<4, 4, 16>(Y, X[k], ldsv + cur, 0, lane);
            }
            if constexpr (ACTS) { if (act_on) store_rows<4>(act_rs, acol, act_stride4, ROW_Y, Y); }
            // views_linear: [feature, enc(dir)] -> V (chunks 14,15,16), 64 units = 2 tiles
            f32x16 V[2];
            load_bias<2>(V, small + SM_BV, half);
            HL_CHUNK_ADVANCE(16 % NCH)
            mma16<2, 8, 16>(V, Y[0], ldsv + cur, 0, lane);
            mma16<2, 8, 16>(V, Y[1], ldsv + cur, 4, lane);
            HL_CHUNK_ADVANCE(17 % NCH)
            mma16<2, 8, 16>(V, Y[2], ldsv + cur, 0, lane);
            mma16<2, 8, 16>(V, Y[3], ldsv + cur, 4, lane);
            HL_CHUNK_ADVANCE(18 % NCH)
            mma16<2, 4, 14>(V, ev, ldsv + cur, 0, lane);
            V[0] = softplus16(V[0]);
            V[1] = softplus16(V[1]);
            if constexpr (ACTS) { if (act_on) store_rows<2>(act_rs, acol, act_stride4, ROW_V, V); }
            const float cr = dot_lane<2>(V, small + SM_RW, half) + small[SM_RB + 0];
            const float cg = dot_lane<2>(V, small + SM_RW + 64, half) + small[SM_RB + 1];
            const float cb = dot_lane<2>(V, small + SM_RW + 128, half) + small[SM_RB + 2];
            HL_CHUNK_ADVANCE(19 % NCH)  // chunk 0 of the next sample

            if constexpr (STORE) {
                // The two halves of a ray hold the same four values: lanes 0-31 store (sigma, r), lanes 32-63 (g, b), 8 bytes each.
                // The store is issued through inline asm on purpose: a compiler-visible VMEM write in this loop makes the
                // waitcnt pass treat the vmcnt queue as mixed read/write and turn the counted waits of the weight prefetch
                // and the plane gathers into vmcnt(0) (measured: 50 ms instead of 38 ms per 128-sample pass).  Hidden stores
                // are safe for the compiler's load waits - an outstanding store can only make a counted wait stricter - and
                // nothing in this kernel reads the records back.
                if (tile * 32 < a.R) {
                    const float2 rec = half ? make_float2(cg, cb) : make_float2(sigma_raw, cr);
                    float *dst = reinterpret_cast<float *>(a.vals_out + (zt_base + 32LL * s)) + 2 * half;
                    asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(dst), "v"(rec) : "memory");
                }
            }
            // ---- alpha compositing  [renderer.py:185-186, 213, 221-229] ----
            // (also in STORE mode, where its result is discarded: dropping it sends the register allocator of this ROCm down a
            //  path with 6x the spills - 1.2 KB of scratch per lane - and a third of the throughput)
            {
                const float dist = (s + 1 < S) ? zn - zc : 1e10f;
                const float alpha = dist == 1e10f ? comp_alpha_last(sigma_raw) : comp_alpha(sigma_raw, dist);
                const float w = alpha * T;
                acc_w += w;
                acc_r += comp_sigmoid(cr) * w;
                acc_g += comp_sigmoid(cg) * w;
                acc_b += comp_sigmoid(cb) * w;
                acc_d += w * zc;
                T *= (1.f - alpha + 1e-7f);
            }
        }
        zc = zn;
    }
#undef HL_CHUNK_ADVANCE

    if constexpr (FULL) {
        if (valid && half == 0 && (!STORE || a.rgb != nullptr)) {
            if (a.flags & HL_RENDER_WHITE_BKGD) {
                const float bg = 1.f - acc_w;
                acc_r += bg; acc_g += bg; acc_b += bg;
            }
            if (a.flags & HL_RENDER_NORMALIZE_DEPTH) {
                acc_d = (acc_d - nr) / (fr - nr + 1e-5f);
                if (a.flags & HL_RENDER_CLAMP_DEPTH) {
                    acc_d = acc_d > 1.f ? 1.f : acc_d;
                    acc_d = acc_d < 0.f ? 0.f : acc_d;
                }
            }
            a.rgb[ray * 3 + 0] = acc_r;
            a.rgb[ray * 3 + 1] = acc_g;
            a.rgb[ray * 3 + 2] = acc_b;
            a.acc[ray] = acc_w;
            a.depth[ray] = acc_d;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_march16: the evaluate pass (FULL + STORE) with fp16 operands / fp32 accumulation on v_mfma_f32_32x32x16_f16 - OPT-IN
// (HL_RENDER_MLP_FP16), never the default.  The accumulator-is-the-next-B-operand identity survives the wider k-step with a permuted k
// order: a 16-wide k-step takes, from each lane half, EIGHT of the accumulator registers that half already holds (registers 8jj..8jj+7 of
// input tile tin: units unit_of(tin, r, half)) - the weights are packed in that order (k_pack_mlp16), so a layer's input is its
// producer's accumulators rounded to fp16 in place, no cross-lane traffic.  128 -> 128 is 32 MFMAs of 32 cycles instead of 256 of 64, and
// all 132 KB of 16-bit weights sit in LDS for the whole launch: no weight ring, no barrier in the sample loop.
// Fragment f of the packed image = [lane 64][8 halfs]; parts in consumption order, fragment (part, j, t) at part_base + j * nt + t:
constexpr int P16_L0 = 0, P16_L1 = 8, P16_L2F = 40, P16_L2H = 48, P16_FEAT = 80, P16_VH = 112, P16_VE = 128, P16_FRAGS = 132;
struct Part16 { int w, ld, col0, kind, nt, nj, base; };
__constant__ Part16 c_parts16[7] = {{0, 27, 0, 0, 4, 2, P16_L0},   {1, 128, 0, 1, 4, 8, P16_L1},   {2, 155, 0, 0, 4, 2, P16_L2F}, {2, 155, 27, 1, 4, 8, P16_L2H},
                                    {3, 128, 0, 1, 4, 8, P16_FEAT}, {4, 155, 0, 1, 2, 8, P16_VH},   {4, 155, 128, 2, 2, 2, P16_VE}};

__global__ void k_pack_mlp16(PackArgs a, unsigned short *out16) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P16_FRAGS * 512) return;
    const int i = idx & 7, lane = (idx >> 3) & 63, f = idx >> 9;
    int pi = 6;
    while (pi > 0 && f < c_parts16[pi].base) --pi;
    const Part16 d = c_parts16[pi];
    const int rel = f - d.base, j = rel / d.nt, t = rel - j * d.nt;
    const int g = lane >> 5, outu = 32 * t + (lane & 31);
    int in = -1;
    if (d.kind == 1) {
        in = unit_of(j >> 1, (j & 1) * 8 + i, g);
    } else {
        const int sidx = j * 8 + i, per = d.kind == 0 ? 15 : 14, k = sidx + per * g;
        if (sidx < per && k < 27) in = k;
    }
    const float v = in >= 0 ? a.w[d.w][outu * d.ld + d.col0 + in] : 0.f;
    const _Float16 h = (_Float16)v;
    out16[idx] = __builtin_bit_cast(unsigned short, h);
}

// softplus of the fp16-operand mode: ln2 * log2(1 + 2^(x log2e)) - no |x| / max split (for x >> 0 the sum rounds to 2^y and the logarithm
// returns y; for x << 0 it returns 0 where the exact value is e^x < 6e-8 - both far below the fp16 rounding the result meets as an MFMA
// operand), the exponent capped at 2^126.  Three packed fp32 operations + v_min + two transcendentals per value instead of four + two.
__device__ __forceinline__ f32x16 softplus16_h(f32x16 v) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        f32x2_ y = f32x2_{v[i], v[i + 1]} * 1.44269504088896341f;
        y[0] = fminf(y[0], 126.f); y[1] = fminf(y[1], 126.f);
        const f32x2_ e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
        const f32x2_ s1 = e + 1.f;
        const f32x2_ l = f32x2_{__builtin_amdgcn_logf(s1[0]), __builtin_amdgcn_logf(s1[1])} * 0.693147180559945309f;
        o[i] = l[0]; o[i + 1] = l[1];
    }
    return o;
}
__device__ __forceinline__ u32x4 cvt_h8(const f32x16 &v, int hi) {   // registers 8hi..8hi+7 -> 8 fp16 (nearest even)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const h2 h = {(_Float16)v[8 * hi + 2 * q], (_Float16)v[8 * hi + 2 * q + 1]};
        o[q] = __builtin_bit_cast(unsigned, h);
    }
    return o;
}
template <int NT>
__device__ __forceinline__ void mma_h(f32x16 (&acc)[NT], const u32x4 b, const u32x4 *__restrict__ frags, int lane) {   // frags: NT consecutive fragments
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, frags[t * 64 + lane]), __builtin_bit_cast(h8, b), acc[t], 0, 0, 0);
}

template <int NWV>
__global__ __launch_bounds__(NWV * 64, NWV / 4) void k_march16(const MarchArgs a, const unsigned short *__restrict__ packed16) {
    extern __shared__ __attribute__((aligned(16))) float lds16[];
    constexpr int NT = NWV * 64;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const long long wg = (long long)xcd * q8 + (xcd < r8 ? xcd : r8) + (blockIdx.x >> 3);
    const long long tile = wg * NWV + (tid >> 6);
    const long long ray = tile * 32 + (lane & 31);
    const bool valid = ray < a.R;
    const long long rc = valid ? ray : a.R - 1;
    const long long tiles_n = (a.R + 31) / 32;
    const long long zt_base = (tile < tiles_n ? tile : tiles_n - 1) * 32 * (long long)a.S + (lane & 31);

    // LDS: [P16_FRAGS fragments of 1 KB][small parameter block, fp32, as k_march's]
    u32x4 *fr = reinterpret_cast<u32x4 *>(lds16);
    const u32x4 *g16 = reinterpret_cast<const u32x4 *>(packed16);
    for (int i = tid; i < P16_FRAGS * 64; i += NT) fr[i] = g16[i];
    float *small = lds16 + P16_FRAGS * 256;
    for (int i = tid; i < SMALL_FLOATS; i += NT) small[i] = a.packed[NCH_FULL * CHUNK_FLOATS + i];

    const float ox = a.rays_o[rc * 3 + 0], oy = a.rays_o[rc * 3 + 1], oz = a.rays_o[rc * 3 + 2];
    const float dx = a.rays_d[rc * 3 + 0], dy = a.rays_d[rc * 3 + 1], dz = a.rays_d[rc * 3 + 2];
    const float nr = a.near[rc], fr_ = a.far[rc];
    const int S = a.S;
    const float offH = (float)(1.0 / (double)a.H);
    const float bmin0 = a.bounds[0], bmin1 = a.bounds[1], bmin2 = a.bounds[2];
    const float bext0 = a.bounds[3] - bmin0, bext1 = a.bounds[4] - bmin1, bext2 = a.bounds[5] - bmin2;

    // view-direction encoding, this half's 14 of the 27 (+1 pad) entries (as k_march)
    f32x16 ev;
    {
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        const float vd[3] = {dx / nrm, dy / nrm, dz / nrm};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float val = 0.f;
            if (s < 14) {
                const int kl = s, kh = s + 14;
                const int jl = (kl - 3) / 3, cl = (kl - 3) % 3, jh = (kh - 3) / 3, ch = (kh - 3) % 3;
                const float argl = kl < 3 ? 0.f : ((jl & 1) ? 1.57079632679489661923f : 0.f) + vd[kl < 3 ? 0 : cl] * (float)(1 << (jl >> 1));
                const float argh = ((jh & 1) ? 1.57079632679489661923f : 0.f) + vd[ch] * (float)(1 << (jh >> 1));
                if (kl < 3) {
                    const float sh = sinf(argh);
                    val = half ? sh : vd[kl];
                } else if (kh >= 27) {
                    const float sl = sinf(argl);
                    val = half ? 0.f : sl;
                } else {
                    val = sinf(half ? argh : argl);
                }
            }
            ev[s] = val;
        }
    }
    const u32x4 bev0 = cvt_h8(ev, 0), bev1 = cvt_h8(ev, 1);

    float zc;
    if (a.z) zc = a.z_tiled ? a.z[zt_base] : a.z[rc * S];
    else zc = nr * (1.f - linspace01(0, S)) + fr_ * linspace01(0, S);
    __syncthreads();

    for (int s = 0; s < S; ++s) {
        float zn = 0.f;
        if (s + 1 < S) {
            if (a.z) zn = a.z_tiled ? a.z[zt_base + 32LL * (s + 1)] : a.z[rc * S + s + 1];
            else { const float t = linspace01(s + 1, S); zn = nr * (1.f - t) + fr_ * t; }
        }
        // ---- tri-plane features of this half (fp32, exactly as k_march)  [renderer.py:502-531] ----
        const float px = ox + dx * zc, py = oy + dy * zc, pz = oz + dz * zc;
        const float nx = 2.f * (px - bmin0) / bext0 - 1.f;
        const float ny = 2.f * (py - bmin1) / bext1 - 1.f;
        const float nz = 2.f * (pz - bmin2) / bext2 - 1.f;
        f32x16 f;
        f[15] = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int qlo = i, qhi = (i + 5 > 8) ? 8 : i + 5;
            const int q = half ? qhi : qlo;
            const int p = half ? qhi / 3 : qlo / 3, g = half ? qhi % 3 : qlo % 3;
            float gu = (p == 2) ? nz : nx;
            float gv = (p == 1) ? nz : ny;
            gu = (g == 1) ? gu + offH : gu;
            gv = (g == 2) ? gv + offH : gv;
            const float ix = ((gu + 1.f) * (float)a.W - 1.f) / 2.f;
            const float iy = ((gv + 1.f) * (float)a.H - 1.f) / 2.f;
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float x1f = x0f + 1.f, y1f = y0f + 1.f;
            float w_nw = (x1f - ix) * (y1f - iy), w_ne = (ix - x0f) * (y1f - iy);
            float w_sw = (x1f - ix) * (iy - y0f), w_se = (ix - x0f) * (iy - y0f);
            const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
            const bool vx0 = (x0 >= 0) & (x0 < a.W), vx1 = (x1 >= 0) & (x1 < a.W);
            const bool vy0 = (y0 >= 0) & (y0 < a.H), vy1 = (y1 >= 0) & (y1 < a.H);
            w_nw = (vx0 & vy0) ? w_nw : 0.f;
            w_ne = (vx1 & vy0) ? w_ne : 0.f;
            w_sw = (vx0 & vy1) ? w_sw : 0.f;
            w_se = (vx1 & vy1) ? w_se : 0.f;
            const int cx0 = min(max(x0, 0), a.W - 1), cx1 = min(max(x1, 0), a.W - 1);
            const int cy0 = min(max(y0, 0), a.H - 1), cy1 = min(max(y1, 0), a.H - 1);
            const float4 *pl = a.planes + (long long)q * a.H * a.W;
            const float4 t_nw = pl[cy0 * a.W + cx0], t_ne = pl[cy0 * a.W + cx1];
            const float4 t_sw = pl[cy1 * a.W + cx0], t_se = pl[cy1 * a.W + cx1];
            const bool live = half ? (i + 5 <= 8) : true;
            const float r0 = t_nw.x * w_nw + t_ne.x * w_ne + t_sw.x * w_sw + t_se.x * w_se;
            const float r1 = t_nw.y * w_nw + t_ne.y * w_ne + t_sw.y * w_sw + t_se.y * w_se;
            const float r2 = t_nw.z * w_nw + t_ne.z * w_ne + t_sw.z * w_sw + t_se.z * w_se;
            f[3 * i + 0] = live ? r0 : 0.f;
            f[3 * i + 1] = live ? r1 : 0.f;
            f[3 * i + 2] = live ? r2 : 0.f;
        }
        const u32x4 bf0 = cvt_h8(f, 0), bf1 = cvt_h8(f, 1);
        // ---- MLP  [renderer.py:134-156] ----
        f32x16 X[4], Y[4];
        load_bias<4>(X, small + SM_B0, half);
        mma_h<4>(X, bf0, fr + (P16_L0 + 0) * 64, lane);
        mma_h<4>(X, bf1, fr + (P16_L0 + 4) * 64, lane);
        load_bias<4>(Y, small + SM_B1, half);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            X[k] = softplus16_h(X[k]);
            mma_h<4>(Y, cvt_h8(X[k], 0), fr + (P16_L1 + (2 * k) * 4) * 64, lane);
            mma_h<4>(Y, cvt_h8(X[k], 1), fr + (P16_L1 + (2 * k + 1) * 4) * 64, lane);
        }
        load_bias<4>(X, small + SM_B2, half);
        mma_h<4>(X, bf0, fr + (P16_L2F + 0) * 64, lane);
        mma_h<4>(X, bf1, fr + (P16_L2F + 4) * 64, lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            Y[k] = softplus16_h(Y[k]);
            mma_h<4>(X, cvt_h8(Y[k], 0), fr + (P16_L2H + (2 * k) * 4) * 64, lane);
            mma_h<4>(X, cvt_h8(Y[k], 1), fr + (P16_L2H + (2 * k + 1) * 4) * 64, lane);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = softplus16_h(X[k]);
        const float sigma_raw = dot_lane<4>(X, small + SM_AW, half) + small[SM_AB];
        load_bias<4>(Y, small + SM_BF, half);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mma_h<4>(Y, cvt_h8(X[k], 0), fr + (P16_FEAT + (2 * k) * 4) * 64, lane);
            mma_h<4>(Y, cvt_h8(X[k], 1), fr + (P16_FEAT + (2 * k + 1) * 4) * 64, lane);
        }
        f32x16 V[2];
        load_bias<2>(V, small + SM_BV, half);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mma_h<2>(V, cvt_h8(Y[k], 0), fr + (P16_VH + (2 * k) * 2) * 64, lane);
            mma_h<2>(V, cvt_h8(Y[k], 1), fr + (P16_VH + (2 * k + 1) * 2) * 64, lane);
        }
        mma_h<2>(V, bev0, fr + (P16_VE + 0) * 64, lane);
        mma_h<2>(V, bev1, fr + (P16_VE + 2) * 64, lane);
        V[0] = softplus16_h(V[0]);
        V[1] = softplus16_h(V[1]);
        const float cr = dot_lane<2>(V, small + SM_RW, half) + small[SM_RB + 0];
        const float cg = dot_lane<2>(V, small + SM_RW + 64, half) + small[SM_RB + 1];
        const float cb = dot_lane<2>(V, small + SM_RW + 128, half) + small[SM_RB + 2];
        if (tile * 32 < a.R) {   // lanes 0-31 store (sigma, r), lanes 32-63 (g, b)
            const float2 rec = half ? make_float2(cg, cb) : make_float2(sigma_raw, cr);
            float *dst = reinterpret_cast<float *>(a.vals_out + (zt_base + 32LL * s)) + 2 * half;
            asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(dst), "v"(rec) : "memory");
        }
        zc = zn;
    }
}

// ---------------------------------------------------------------------------------------------
// k_march_b3 (round 4): the evaluate pass (FULL + STORE) with the fp32 products formed on the 16-bit matrix pipe WITHOUT leaving fp32
// tolerance - both operands of every product are split EXACTLY into three bf16 planes (x = x0 + x1 + x2: 3 x 8 significand bits hold all
// 24 of an fp32 value; the weights once at pack time, an activation fragment in registers right where k_march16 rounds it to fp16) and
// the six partial products of weight >= 2^-16 are accumulated in fp32 on v_mfma_f32_32x32x16_bf16; the three dropped terms are below
// 2^-22 |a b| in the worst case (truncated activation planes against nearest-even weight planes), far less on average - the order of
// the rounding of one fp32 fma.  Same fragment order as k_march16 (the accumulator-is-the-next-B-operand identity under the permuted k
// order), same fp32 gather / softplus / heads as k_march.
//   fp32 kernel : 1 044 MFMAs of 64 cycles per sample and wave, and every VALU instruction is ADDED to them (the fp32 MFMA runs on the
//                 SIMD's fp32 lanes: profiles/r03_microbench_mfma_fill.txt)                                  -> 84k cycles, 0.78 of the peak
//   this kernel :   792 MFMAs of 32 cycles, and ~5 VALU issues hide behind each (profiles/r04_microbench_mfma16_mix.txt)
// The three planes are 396 KB: they stream from L2 through an LDS ring, the chunk pair after next staged in registers - the scheme of
// k_march.  Chunk = 4 fragment positions x 3 planes = the weights of one 8-wide k-group for four output tiles (12 KB).  4 waves per
// workgroup = one per SIMD (the 512-entry register budget).
constexpr int B3_POS = 4, B3_NCH = P16_FRAGS / B3_POS, B3_CH_U4 = B3_POS * 3 * 64;   // positions per chunk, chunks, u32x4 per chunk (12 KB)
static_assert(P16_FRAGS % B3_POS == 0, "chunking");
constexpr size_t B3_BYTES = (size_t)P16_FRAGS * 3 * 1024;

__device__ __forceinline__ unsigned short bf16_rne(float v) {   // nearest even (the values here are finite)
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__global__ void k_pack_mlp_b3(PackArgs a, unsigned short *out) {   // [position][plane][lane 64][8 bf16]
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P16_FRAGS * 3 * 512) return;
    const int i = idx & 7, lane = (idx >> 3) & 63, rest = idx >> 9, plane = rest % 3, f = rest / 3;
    int pi = 6;
    while (pi > 0 && f < c_parts16[pi].base) --pi;
    const Part16 d = c_parts16[pi];
    const int rel = f - d.base, j = rel / d.nt, t = rel - j * d.nt;
    const int g = lane >> 5, outu = 32 * t + (lane & 31);
    int in = -1;
    if (d.kind == 1) {
        in = unit_of(j >> 1, (j & 1) * 8 + i, g);
    } else {
        const int sidx = j * 8 + i, per = d.kind == 0 ? 15 : 14, k = sidx + per * g;
        if (sidx < per && k < 27) in = k;
    }
    float v = in >= 0 ? a.w[d.w][outu * d.ld + d.col0 + in] : 0.f;
    unsigned short h = 0;
    for (int p = 0; p <= plane; ++p) {
        h = bf16_rne(v);
        v -= __builtin_bit_cast(float, (unsigned)h << 16);       // exact
    }
    out[idx] = h;
}

// the same layout with TWO fp16 planes, nearest even at both levels (fp16x2 products, k_march_plw<2>): [position][plane][lane 64][8 fp16]
__global__ void k_pack_mlp_h2(PackArgs a, unsigned short *out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P16_FRAGS * 2 * 512) return;
    const int i = idx & 7, lane = (idx >> 3) & 63, rest = idx >> 9, plane = rest % 2, f = rest / 2;
    int pi = 6;
    while (pi > 0 && f < c_parts16[pi].base) --pi;
    const Part16 d = c_parts16[pi];
    const int rel = f - d.base, j = rel / d.nt, t = rel - j * d.nt;
    const int g = lane >> 5, outu = 32 * t + (lane & 31);
    int in = -1;
    if (d.kind == 1) {
        in = unit_of(j >> 1, (j & 1) * 8 + i, g);
    } else {
        const int sidx = j * 8 + i, per = d.kind == 0 ? 15 : 14, k = sidx + per * g;
        if (sidx < per && k < 27) in = k;
    }
    float v = in >= 0 ? a.w[d.w][outu * d.ld + d.col0 + in] : 0.f;
#ifndef HL_H2_NO_LOG2
    // log2-domain softplus (k_march_plw<2>, see softplus_l2): a layer in front of a softplus produces log2(e) x its pre-activation, a layer behind one
    // consumes log2-unit activations (x ln 2) - for hidden -> hidden layers the two cancel and the planes are those of the unscaled weights
    constexpr float PSC[7] = {1.44269504088896341f, 1.f, 1.44269504088896341f, 1.f, 0.693147180559945309f, 1.44269504088896341f, 1.44269504088896341f};
    v *= PSC[pi];
    v *= a.out[NCH_FULL * CHUNK_FLOATS + SM_SC + d.w];          // the layer's power of two (k_mlp_scales_h2): exact
#endif
    _Float16 h = (_Float16)v;                                   // nearest even
    if (plane == 1) h = (_Float16)(v - (float)h);               // (the residual is exact in fp32)
    out[idx] = __builtin_bit_cast(unsigned short, h);
}

// registers 8hi..8hi+7 of an accumulator tile -> three planes of 8 bf16 (v = p0 + p1 + p2 exactly)
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void split_b3(const f32x16 &v, int hi, u32x4 (&pl)[3]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x = v[8 * hi + 2 * q], y = v[8 * hi + 2 * q + 1];
        const unsigned p0 = cvt_pk_bf16(x, y);
        x -= __builtin_bit_cast(float, p0 << 16); y -= __builtin_bit_cast(float, p0 & 0xffff0000u);
        const unsigned p1 = cvt_pk_bf16(x, y);
        x -= __builtin_bit_cast(float, p1 << 16); y -= __builtin_bit_cast(float, p1 & 0xffff0000u);
        pl[0][q] = p0; pl[1][q] = p1; pl[2][q] = cvt_pk_bf16(x, y);
    }
}
// ---------------------------------------------------------------------------------------------
// The instruction stream of k_march_b3 is laid out by hand.  Left to itself the compiler keeps the VALU work of a layer (softplus, the three-way
// split) outside the MFMA groups, where nothing overlaps it (ablations in profiles/r04_render_b3_ablations.md: everything adds up).  Measured
// with one wave per SIMD (profiles/r04_microbench_mfma16_mix.txt): behind a v_mfma_f32_32x32x16_bf16 five plain VALU issues are free, a gap
// then costs ~8 + 4.8 cycles per plain instruction, and v_exp / v_log / v_cvt_pk_bf16_f32 / v_accvgpr_read count double.  Hence:
//   * every MFMA of a chunk is followed by a fixed slice (~5 plain-instruction equivalents) of the work that prepares the NEXT chunk's B
//     operand: softplus + split of a pair of values = six slices, a raw split = three;
//   * the split truncates instead of rounding (v_and + v_perm, plain rate; still exact: 3 x 8 bits) and max(x, 0) is a v_max_i32;
//   * the next chunk's weight fragments are read from LDS one chunk ahead (one ds_read_b128 behind each of the first twelve MFMAs);
//   * where a chunk's operand depends on the layer that is just finishing, that layer's last chunk runs tile 0 first and prepares the
//     operand behind the MFMAs of tiles 1..3;
//   * the ring holds three slots of two chunks (24 KB): one barrier per 48 MFMAs.
struct B3Op { u32x4 p[3]; };
struct B3Tmp { float x, y, ex, ey; unsigned hx, hy; };
template <int N, class F, int... I>
__device__ __forceinline__ void b3_seq_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void b3_seq(F &&f) { b3_seq_impl<N>(f, std::make_integer_sequence<int, N>{}); }
__device__ __forceinline__ float b3_max0(float x) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }   // v_max_i32: max(x, +0) for every non-NaN x
__device__ __forceinline__ unsigned b3_hi(float x) { return __builtin_bit_cast(unsigned, x) & 0xffff0000u; }
__device__ __forceinline__ unsigned b3_pack(float x, float y) { return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, y), __builtin_bit_cast(unsigned, x), 0x07060302u); }
// Preparation of pair Q (registers 8H + 2Q, +1 of `src`) of operand `o`, slice T.  KIND 0: raw split, T = 0..2; KIND 1: softplus, then
// split, T = 0..5; KIND 2: as 1, and the softplus values also feed the density head (asum += value * aw[register]).
template <int KIND, int H, int Q, int T>
__device__ __forceinline__ void b3_prep(const f32x16 &src, B3Op &o, B3Tmp &t, float &asum, const float *__restrict__ aw) {
    constexpr int r = 8 * H + 2 * Q;
    if constexpr (KIND == 0) {
        if constexpr (T == 0) { t.x = src[r]; t.y = src[r + 1]; t.hx = b3_hi(t.x); }
        else if constexpr (T == 1) {
            t.hy = b3_hi(t.y); o.p[0][Q] = b3_pack(t.x, t.y);
            t.x -= __builtin_bit_cast(float, t.hx); t.y -= __builtin_bit_cast(float, t.hy); t.hx = b3_hi(t.x);
        } else {
            t.hy = b3_hi(t.y); o.p[1][Q] = b3_pack(t.x, t.y);
            t.x -= __builtin_bit_cast(float, t.hx); t.y -= __builtin_bit_cast(float, t.hy); o.p[2][Q] = b3_pack(t.x, t.y);
        }
    } else {
        if constexpr (T == 0) {
            t.x = src[r]; t.y = src[r + 1];
            t.ex = -1.44269504088896341f * fabsf(t.x); t.ey = -1.44269504088896341f * fabsf(t.y);
        } else if constexpr (T == 1) {
            t.ex = __builtin_amdgcn_exp2f(t.ex); t.ey = __builtin_amdgcn_exp2f(t.ey); t.ex = 1.f + t.ex;
        } else if constexpr (T == 2) {
            t.ey = 1.f + t.ey; t.ex = __builtin_amdgcn_logf(t.ex); t.ey = __builtin_amdgcn_logf(t.ey);
        } else if constexpr (T == 3) {
            t.x = fmaf(0.693147180559945309f, t.ex, b3_max0(t.x));
            t.y = fmaf(0.693147180559945309f, t.ey, b3_max0(t.y));
            if constexpr (KIND == 2) { const float2 w2 = *reinterpret_cast<const float2 *>(aw + r); asum = fmaf(t.x, w2.x, asum); asum = fmaf(t.y, w2.y, asum); }
            t.hx = b3_hi(t.x);
        } else if constexpr (T == 4) {
            t.hy = b3_hi(t.y); o.p[0][Q] = b3_pack(t.x, t.y);
            t.x -= __builtin_bit_cast(float, t.hx); t.y -= __builtin_bit_cast(float, t.hy); t.hx = b3_hi(t.x);
        } else {
            t.hy = b3_hi(t.y); o.p[1][Q] = b3_pack(t.x, t.y);
            t.x -= __builtin_bit_cast(float, t.hx); t.y -= __builtin_bit_cast(float, t.hy); o.p[2][Q] = b3_pack(t.x, t.y);
        }
    }
}
// slice ST of one operand: 4 pairs x (3 | 6) slices
template <int KIND, int H, int ST>
__device__ __forceinline__ void b3_prep_step(const f32x16 &src, B3Op &o, B3Tmp &t, float &asum, const float *__restrict__ aw) {
    constexpr int NS = KIND == 0 ? 3 : 6;
    if constexpr (ST < 4 * NS) b3_prep<KIND, H, ST / NS, ST % NS>(src, o, t, asum, aw);
}
// The 24 MFMAs of a chunk.  NT2 == 4: one k-group (operand oa) x four output tiles (positions 0..3); NT2 == 2: two k-groups (oa, ob) x two
// tiles (positions 0,1 | 2,3).  TM (tile-major, NT2 == 4): tile 0 completes before tile 1 starts, ...  after(idx) runs behind MFMA idx.
template <int NT2, bool TM, class ACC, class F>
__device__ __forceinline__ void b3_chunk(ACC &acc, const B3Op &oa, const B3Op &ob, const u32x4 (&w)[12], F &&after) {
    constexpr int PW[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
    b3_seq<24>([&](auto ic) {
        constexpr int idx = decltype(ic)::value;
        constexpr int i = NT2 == 4 ? (TM ? idx % 6 : idx / 4) : (idx % 12) / 2;
        constexpr int t = NT2 == 4 ? (TM ? idx / 6 : idx % 4) : (idx & 1);
        constexpr int q = NT2 == 4 ? t : 2 * (idx / 12) + t;
        const B3Op &b = (NT2 == 2 && idx >= 12) ? ob : oa;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[q * 3 + PW[i]]), __builtin_bit_cast(bf16x8, b.p[PB[i]]), acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        after(ic);
        __builtin_amdgcn_sched_barrier(0);
    });
}
constexpr int B3R_NPAIR = (B3_NCH + 1) / 2, B3R_SLOT_U4 = 2 * B3_CH_U4;                     // 17 chunk pairs per sample; 1536 u32x4 = 24 KB per ring slot
constexpr size_t B3R_LDS = (size_t)3 * B3R_SLOT_U4 * 16 + SMALL_FLOATS * sizeof(float);

template <int ABL>
__global__ __launch_bounds__(256, 1) void k_march_b3(const MarchArgs a, const unsigned short *__restrict__ packed_b3) {
    extern __shared__ __attribute__((aligned(16))) float ldsb[];
    constexpr int NT = 256, NST = B3R_SLOT_U4 / NT;   // threads; u32x4 per thread and chunk pair (6)
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const long long wg = (long long)xcd * q8 + (xcd < r8 ? xcd : r8) + (blockIdx.x >> 3);
    const long long tile = wg * 4 + (tid >> 6);
    const long long ray = tile * 32 + (lane & 31);
    const bool valid = ray < a.R;
    const long long rc = valid ? ray : a.R - 1;
    const long long tiles_n = (a.R + 31) / 32;
    const long long zt_base = (tile < tiles_n ? tile : tiles_n - 1) * 32 * (long long)a.S + (lane & 31);

    u32x4 *ring = reinterpret_cast<u32x4 *>(ldsb);
    float *small = ldsb + 3 * B3R_SLOT_U4 * 4;
    for (int i = tid; i < SMALL_FLOATS; i += NT) small[i] = a.packed[NCH_FULL * CHUNK_FLOATS + i];
    // the image is 33 chunks; the 34th (second half of the last pair) lies beyond the buffer's range and loads as zeros - it is never multiplied
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)packed_b3, (short)0, (int)B3_BYTES, 0x00020000);
    const int wv = tid * 16;
    auto ldw = [&](int u4_index) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(rsw, wv, u4_index * 16, 0); };
#pragma unroll
    for (int q = 0; q < 2 * NST; ++q) ring[q * NT + tid] = ldw(q * NT);                       // pairs 0, 1 -> slots 0, 1
    constexpr bool DMA = (ABL & 2048) != 0;   // developer A/B: the ring filled by LDS-DMA (buffer_load ... lds) instead of loads + ds_write_b128
    u32x4 st[NST];
    if constexpr (DMA) {
#pragma unroll
        for (int q = 0; q < NST; ++q) ring[2 * B3R_SLOT_U4 + q * NT + tid] = ldw(2 * B3R_SLOT_U4 + q * NT);   // pair 2 -> slot 2
    } else {
#pragma unroll
        for (int q = 0; q < NST; ++q) st[q] = ldw(2 * B3R_SLOT_U4 + q * NT);                  // pair 2 staged
    }
    int pslot = 0;                                                                            // ring slot (0..2) of the pair being multiplied

    const float ox = a.rays_o[rc * 3 + 0], oy = a.rays_o[rc * 3 + 1], oz = a.rays_o[rc * 3 + 2];
    const float dx = a.rays_d[rc * 3 + 0], dy = a.rays_d[rc * 3 + 1], dz = a.rays_d[rc * 3 + 2];
    const float nr = a.near[rc], fr_ = a.far[rc];
    const int S = a.S;
    const float offH = (float)(1.0 / (double)a.H);
    const float bmin0 = a.bounds[0], bmin1 = a.bounds[1], bmin2 = a.bounds[2];
    const float bext0 = a.bounds[3] - bmin0, bext1 = a.bounds[4] - bmin1, bext2 = a.bounds[5] - bmin2;

    B3Op bev0, bev1;
    {
        f32x16 ev;
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        const float vd[3] = {dx / nrm, dy / nrm, dz / nrm};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float val = 0.f;
            if (s < 14) {
                const int kl = s, kh = s + 14;
                const int jl = (kl - 3) / 3, cl = (kl - 3) % 3, jh = (kh - 3) / 3, ch = (kh - 3) % 3;
                const float argl = kl < 3 ? 0.f : ((jl & 1) ? 1.57079632679489661923f : 0.f) + vd[kl < 3 ? 0 : cl] * (float)(1 << (jl >> 1));
                const float argh = ((jh & 1) ? 1.57079632679489661923f : 0.f) + vd[ch] * (float)(1 << (jh >> 1));
                if (kl < 3) {
                    const float sh = sinf(argh);
                    val = half ? sh : vd[kl];
                } else if (kh >= 27) {
                    const float sl = sinf(argl);
                    val = half ? 0.f : sl;
                } else {
                    val = sinf(half ? argh : argl);
                }
            }
            ev[s] = val;
        }
        split_b3(ev, 0, bev0.p);
        split_b3(ev, 1, bev1.p);
    }

    float zc;
    if (a.z) zc = a.z_tiled ? a.z[zt_base] : a.z[rc * S];
    else zc = nr * (1.f - linspace01(0, S)) + fr_ * linspace01(0, S);
    __syncthreads();
    u32x4 w[12];                                                                              // fragments of the current chunk
#pragma unroll
    for (int i = 0; i < 12; ++i) w[i] = ring[i * 64 + lane];

    // Ring while pair P (chunks 2P, 2P+1) is multiplied: its slot and the next pair's are complete and visible; the slot of pair P-1 takes
    // the staged pair P+2 right after the barrier that opens P, the staging registers then receive pair P+3.  A chunk's fragments are read
    // into registers behind the MFMAs of the chunk before it.
    // run(g, ...): [g even: barrier, ring write, loads] the 24 MFMAs of chunk g with their slices and the fragment reads of chunk g+1.
    const float *aw = small + SM_AW;
    auto run = [&](auto gc, auto nt2c, auto tmc, auto &acc, const B3Op &oa, const B3Op &ob, auto &&slice) {
        constexpr int g = decltype(gc)::value, NT2 = decltype(nt2c)::value;
        constexpr bool TM = decltype(tmc)::value;
        constexpr bool last_of_pair = (g & 1) || g == B3_NCH - 1;
        const int s_cur = pslot * B3R_SLOT_U4, s_nxt = (pslot == 2 ? 0 : pslot + 1) * B3R_SLOT_U4, s_old = (pslot == 0 ? 2 : pslot - 1) * B3R_SLOT_U4;
        constexpr bool ring_here = !(ABL & 4) && (g & 1) == 0;       // the first chunk of a pair carries the ring traffic
        if constexpr (ring_here && DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the pair requested one pair ago have landed
        if constexpr (ring_here && !(ABL & 64)) __syncthreads();      // (64: timing only - no barrier; 128: no ring traffic; 512: no LDS writes; 1024: no loads)
        const u32x4 *nxt = ring + ((ABL & 4) ? 0 : (last_of_pair ? s_nxt : s_cur + B3_CH_U4));
        u32x4 wn[12];
        b3_chunk<NT2, TM>(acc, oa, ob, w, [&](auto ic) {
            constexpr int idx = decltype(ic)::value;
            if constexpr (idx < 12) wn[idx] = nxt[idx * 64 + lane];
            // ring traffic one instruction per gap (in a burst behind the barrier each ds_write_b128 cost ~100 cycles: profiles/r04_render_b3_ablations.md)
            if constexpr (DMA) {
                if constexpr (ring_here && !(ABL & 128) && idx >= 12 && idx < 12 + NST)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void *)(ring + s_old + (idx - 12) * NT + (tid & ~63)), 16, lane * 16,
                                                             (((g / 2 + 2) % B3R_NPAIR) * B3R_SLOT_U4 + (idx - 12) * NT) * 16 + (tid & ~63) * 16, 0, 0);
            } else {
                if constexpr (ring_here && !(ABL & 128) && idx >= 12 && idx < 12 + NST) {
                    if constexpr (!(ABL & 512)) ring[s_old + (idx - 12) * NT + tid] = st[idx - 12];
                    else asm volatile("" ::"v"(st[idx - 12]));
                }
                if constexpr (ring_here && !(ABL & 128) && !(ABL & 1024) && idx >= 12 + NST && idx < 12 + 2 * NST)
                    st[idx - 12 - NST] = ldw(((g / 2 + 3) % B3R_NPAIR) * B3R_SLOT_U4 + (idx - 12 - NST) * NT);
            }
            if constexpr (!(ABL & 256)) slice(ic);                   // (256: timing only - no operand preparation behind the MFMAs)
        });
#pragma unroll
        for (int i = 0; i < 12; ++i) w[i] = wn[i];
        if constexpr (last_of_pair) pslot = pslot == 2 ? 0 : pslot + 1;
    };
    using I0 = std::integral_constant<int, 0>;
    auto none = [](auto) {};
    using C4 = std::integral_constant<int, 4>;
    using C2 = std::integral_constant<int, 2>;
    using TMy = std::true_type;
    using TMn = std::false_type;

    for (int s = 0; s < S; ++s) {
        float zn = 0.f;
        if (s + 1 < S) {
            if (a.z) zn = a.z_tiled ? a.z[zt_base + 32LL * (s + 1)] : a.z[rc * S + s + 1];
            else { const float t = linspace01(s + 1, S); zn = nr * (1.f - t) + fr_ * t; }
        }
        // ---- tri-plane features of this half (fp32, exactly as k_march)  [renderer.py:502-531] ----
        const float px = ox + dx * zc, py = oy + dy * zc, pz = oz + dz * zc;
        const float nx = 2.f * (px - bmin0) / bext0 - 1.f;
        const float ny = 2.f * (py - bmin1) / bext1 - 1.f;
        const float nz = 2.f * (pz - bmin2) / bext2 - 1.f;
        f32x16 f;
        f[15] = 0.f;
        if constexpr (ABL & 1) {
#pragma unroll
            for (int i = 0; i < 15; ++i) f[i] = nx * (float)i + ny;
        } else
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int qlo = i, qhi = (i + 5 > 8) ? 8 : i + 5;
            const int q = half ? qhi : qlo;
            const int p = half ? qhi / 3 : qlo / 3, g = half ? qhi % 3 : qlo % 3;
            float gu = (p == 2) ? nz : nx;
            float gv = (p == 1) ? nz : ny;
            gu = (g == 1) ? gu + offH : gu;
            gv = (g == 2) ? gv + offH : gv;
            const float ix = ((gu + 1.f) * (float)a.W - 1.f) / 2.f;
            const float iy = ((gv + 1.f) * (float)a.H - 1.f) / 2.f;
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float x1f = x0f + 1.f, y1f = y0f + 1.f;
            float w_nw = (x1f - ix) * (y1f - iy), w_ne = (ix - x0f) * (y1f - iy);
            float w_sw = (x1f - ix) * (iy - y0f), w_se = (ix - x0f) * (iy - y0f);
            const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
            const bool vx0 = (x0 >= 0) & (x0 < a.W), vx1 = (x1 >= 0) & (x1 < a.W);
            const bool vy0 = (y0 >= 0) & (y0 < a.H), vy1 = (y1 >= 0) & (y1 < a.H);
            w_nw = (vx0 & vy0) ? w_nw : 0.f;
            w_ne = (vx1 & vy0) ? w_ne : 0.f;
            w_sw = (vx0 & vy1) ? w_sw : 0.f;
            w_se = (vx1 & vy1) ? w_se : 0.f;
            const int cx0 = min(max(x0, 0), a.W - 1), cx1 = min(max(x1, 0), a.W - 1);
            const int cy0 = min(max(y0, 0), a.H - 1), cy1 = min(max(y1, 0), a.H - 1);
            const float4 *pl = a.planes + (long long)q * a.H * a.W;
            const float4 t_nw = pl[cy0 * a.W + cx0], t_ne = pl[cy0 * a.W + cx1];
            const float4 t_sw = pl[cy1 * a.W + cx0], t_se = pl[cy1 * a.W + cx1];
            const bool live = half ? (i + 5 <= 8) : true;
            const float r0 = t_nw.x * w_nw + t_ne.x * w_ne + t_sw.x * w_sw + t_se.x * w_se;
            const float r1 = t_nw.y * w_nw + t_ne.y * w_ne + t_sw.y * w_sw + t_se.y * w_se;
            const float r2 = t_nw.z * w_nw + t_ne.z * w_ne + t_sw.z * w_sw + t_se.z * w_se;
            f[3 * i + 0] = live ? r0 : 0.f;
            f[3 * i + 1] = live ? r1 : 0.f;
            f[3 * i + 2] = live ? r2 : 0.f;
        }
        B3Op bf0, bf1, oa, ob;                                     // feature operands; the operands in flight
        if constexpr (ABL & 256) { oa = bev0; ob = bev1; bf1 = bev0; }
        B3Tmp tm;
        float asum = 0.f, dummy = 0.f;
        b3_seq<12>([&](auto ic) { b3_prep_step<0, 0, decltype(ic)::value>(f, bf0, tm, dummy, aw); });
        // ---- MLP  [renderer.py:134-156]: chunk g = fragment positions 4g .. 4g+3 ----
        f32x16 X[4], Y[4];
        load_bias<4>(X, small + SM_B0, half);
        // L0 (chunks 0, 1).  Behind chunk 0: the split of the features' second half; chunk 1 tile-major, behind tiles 1..3: X[0] half 0
        run(I0{}, C4{}, TMn{}, X, bf0, bf0, [&](auto ic) { b3_prep_step<0, 1, decltype(ic)::value>(f, bf1, tm, dummy, aw); });
        run(std::integral_constant<int, 1>{}, C4{}, TMy{}, X, bf1, bf1, [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i >= 6) {   // 24 slices behind 18 MFMAs: 4 slices per 3 gaps
                constexpr int lo = (4 * (i - 6)) / 3, hi = (4 * (i - 5)) / 3;
                b3_seq<hi - lo>([&](auto uc) { b3_prep_step<1, 0, lo + decltype(uc)::value>(X[0], oa, tm, dummy, aw); });
            }
        });
        load_bias<4>(Y, small + SM_B1, half);
        // L1 (chunks 2..9): chunk 2 + 2k + h takes softplus(X[k]) half h; the operand of the next chunk is prepared behind this one's MFMAs
        b3_seq<8>([&](auto jc) {
            constexpr int j = decltype(jc)::value, k = j >> 1, h = j & 1;      // this chunk: (k, h); next: (k + h, h ^ 1)
            constexpr int kn = k + h, hn = h ^ 1;
            if constexpr (h == 0) run(std::integral_constant<int, 2 + j>{}, C4{}, TMn{}, Y, oa, oa, [&](auto ic) { b3_prep_step<1, hn, decltype(ic)::value>(X[kn < 4 ? kn : 3], ob, tm, dummy, aw); });
            else if constexpr (kn < 4) run(std::integral_constant<int, 2 + j>{}, C4{}, TMn{}, Y, ob, ob, [&](auto ic) { b3_prep_step<1, hn, decltype(ic)::value>(X[kn < 4 ? kn : 3], oa, tm, dummy, aw); });
            else run(std::integral_constant<int, 2 + j>{}, C4{}, TMn{}, Y, ob, ob, none);
        });
        // L2, feature part (chunks 10, 11); behind chunk 11: softplus(Y[0]) half 0
        load_bias<4>(X, small + SM_B2, half);
        run(std::integral_constant<int, 10>{}, C4{}, TMn{}, X, bf0, bf0, none);
        run(std::integral_constant<int, 11>{}, C4{}, TMn{}, X, bf1, bf1, [&](auto ic) { b3_prep_step<1, 0, decltype(ic)::value>(Y[0], oa, tm, dummy, aw); });
        // L2, hidden part (chunks 12..19); the last chunk tile-major with softplus(X[0]) half 0 (-> density head + feature_linear) behind tiles 1..3
        b3_seq<8>([&](auto jc) {
            constexpr int j = decltype(jc)::value, k = j >> 1, h = j & 1;
            constexpr int kn = k + h, hn = h ^ 1;
            if constexpr (h == 0) run(std::integral_constant<int, 12 + j>{}, C4{}, TMn{}, X, oa, oa, [&](auto ic) { b3_prep_step<1, hn, decltype(ic)::value>(Y[kn < 4 ? kn : 3], ob, tm, dummy, aw); });
            else if constexpr (kn < 4) run(std::integral_constant<int, 12 + j>{}, C4{}, TMn{}, X, ob, ob, [&](auto ic) { b3_prep_step<1, hn, decltype(ic)::value>(Y[kn < 4 ? kn : 3], oa, tm, dummy, aw); });
            else run(std::integral_constant<int, 12 + j>{}, C4{}, TMy{}, X, ob, ob, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i >= 6) {
                    constexpr int lo = (4 * (i - 6)) / 3, hi = (4 * (i - 5)) / 3;
                    b3_seq<hi - lo>([&](auto uc) { b3_prep_step<2, 0, lo + decltype(uc)::value>(X[0], oa, tm, asum, aw + (0 * 2 + half) * 16); });
                }
            });
        });
        // feature_linear (chunks 20..27): softplus(X[k]) half h (the density head rides along); the last chunk tile-major with the raw split of
        // Y[0] (both halves: the two k-groups of chunk 28) behind tiles 1..3
        load_bias<4>(Y, small + SM_BF, half);
        B3Op oc, od;
        if constexpr (ABL & 256) { oc = bev0; od = bev1; }
        b3_seq<8>([&](auto jc) {
            constexpr int j = decltype(jc)::value, k = j >> 1, h = j & 1;
            constexpr int kn = k + h, hn = h ^ 1;
            if constexpr (h == 0) run(std::integral_constant<int, 20 + j>{}, C4{}, TMn{}, Y, oa, oa, [&](auto ic) { b3_prep_step<2, hn, decltype(ic)::value>(X[kn < 4 ? kn : 3], ob, tm, asum, aw + ((kn < 4 ? kn : 3) * 2 + half) * 16); });
            else if constexpr (kn < 4) run(std::integral_constant<int, 20 + j>{}, C4{}, TMn{}, Y, ob, ob, [&](auto ic) { b3_prep_step<2, hn, decltype(ic)::value>(X[kn < 4 ? kn : 3], oa, tm, asum, aw + ((kn < 4 ? kn : 3) * 2 + half) * 16); });
            else run(std::integral_constant<int, 20 + j>{}, C4{}, TMy{}, Y, ob, ob, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i >= 6) {   // 2 x 12 slices behind 18 MFMAs
                    constexpr int lo = (4 * (i - 6)) / 3, hi = (4 * (i - 5)) / 3;
                    b3_seq<hi - lo>([&](auto uc) {
                        constexpr int st_ = lo + decltype(uc)::value;
                        if constexpr (st_ < 12) b3_prep_step<0, 0, st_>(Y[0], oc, tm, dummy, aw);
                        else b3_prep_step<0, 1, st_ - 12>(Y[0], od, tm, dummy, aw);
                    });
                }
            });
        });
        const float sigma_raw = (asum + __shfl_xor(asum, 32)) + small[SM_AB];
        // views_linear, feature part (chunks 28..31: two k-groups x two tiles each), then the direction encoding (chunk 32)
        f32x16 V[2];
        load_bias<2>(V, small + SM_BV, half);
        b3_seq<4>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            auto prep2 = [&](auto ic) {   // the two operands of chunk 28 + k + 1: 2 x 12 slices behind 24 MFMAs
                constexpr int i = decltype(ic)::value;
                if constexpr (k < 3) {
                    B3Op &na = (k & 1) ? oc : oa, &nb = (k & 1) ? od : ob;
                    if constexpr (i < 12) b3_prep_step<0, 0, i>(Y[k + 1 < 4 ? k + 1 : 3], na, tm, dummy, aw);
                    else b3_prep_step<0, 1, i - 12>(Y[k + 1 < 4 ? k + 1 : 3], nb, tm, dummy, aw);
                }
            };
            if constexpr ((k & 1) == 0) run(std::integral_constant<int, 28 + k>{}, C2{}, TMn{}, V, oc, od, prep2);
            else run(std::integral_constant<int, 28 + k>{}, C2{}, TMn{}, V, oa, ob, prep2);
        });
        run(std::integral_constant<int, 32>{}, C2{}, TMn{}, V, bev0, bev1, none);
        V[0] = softplus16(V[0]);
        V[1] = softplus16(V[1]);
        const float cr = dot_lane<2>(V, small + SM_RW, half) + small[SM_RB + 0];
        const float cg = dot_lane<2>(V, small + SM_RW + 64, half) + small[SM_RB + 1];
        const float cb = dot_lane<2>(V, small + SM_RW + 128, half) + small[SM_RB + 2];
        if (tile * 32 < a.R) {   // lanes 0-31 store (sigma, r), lanes 32-63 (g, b); hidden store: see k_march
            const float2 rec = half ? make_float2(cg, cb) : make_float2(sigma_raw, cr);
            float *dst = reinterpret_cast<float *>(a.vals_out + (zt_base + 32LL * s)) + 2 * half;
            asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(dst), "v"(rec) : "memory");
        }
        zc = zn;
    }
}

// ---------------------------------------------------------------------------------------------
// k_march_b3w: the same arithmetic with EIGHT waves per workgroup (two per SIMD, 256 registers each), compiler-scheduled.  The partner wave
// covers what one wave per SIMD has to schedule by hand (LDS and L2 latencies, the VALU phases between MFMA groups), and 256 rays share every
// weight chunk that goes through LDS - half the ring traffic per ray, which is what bounds k_march_b3 (profiles/r04_render_b3_ablations.md).
constexpr int B3W_NPAIR = (B3_NCH - 1) / 2;                                               // 16 chunk pairs per sample: chunk 32 (direction encoding) runs once per ray
// NPL = operand planes: 3 = bf16x3 (six partial products), 2 = fp16x2 (three partial products, below)
template <int NPL> constexpr int PLW_CH_U4 = B3_POS * NPL * 64;            // u32x4 per chunk (4 fragment positions x NPL planes x 64 lanes)
template <int NPL> constexpr int PLW_SLOT_U4 = 2 * PLW_CH_U4<NPL>;         // a ring slot = a PAIR of chunks (24 KB / 16 KB)
template <int NPL> constexpr size_t PLW_BYTES = (size_t)P16_FRAGS * NPL * 1024;
template <int NPL> constexpr size_t PLW_LDS = (size_t)2 * PLW_SLOT_U4<NPL> * 16 + SMALL_FLOATS * sizeof(float) + (size_t)8 * 512 * 16;
template <int NPL> constexpr size_t PLW_LDS_FUSE = PLW_LDS<NPL> + (size_t)8 * 4 * 64 * 16;      // (HL_FUSE_LQ: + the gathered coarse records, [wave 8][4][64 lanes] x 16 bytes)
constexpr size_t B3W_LDS = PLW_LDS<3>;
template <int NT>
__device__ __forceinline__ void load_bias_global(f32x16 (&acc)[NT], const float *__restrict__ tbl, int half) {   // load_bias from the packed image in global memory
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = tbl[(t * 2 + half) * 16 + r];
}
__device__ __forceinline__ void split_b3t(const f32x16 &v, int hi, u32x4 (&pl)[3]) {   // exact three-way split by truncation: v_and / v_perm / v_sub, all plain rate
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x = v[8 * hi + 2 * q], y = v[8 * hi + 2 * q + 1];
        pl[0][q] = b3_pack(x, y);
        x -= __builtin_bit_cast(float, b3_hi(x)); y -= __builtin_bit_cast(float, b3_hi(y));
        pl[1][q] = b3_pack(x, y);
        x -= __builtin_bit_cast(float, b3_hi(x)); y -= __builtin_bit_cast(float, b3_hi(y));
        pl[2][q] = b3_pack(x, y);
    }
}
// fp16x2 kernel (round 5): softplus in the log2 DOMAIN.  With x' = log2(e) x arriving from the matrix pipe (scaled weights, k_pack_mlp_h2) and the consumers
// taking y' = softplus(x) / ln 2 (their weights x ln 2), a unit costs v_exp_f32, v_add_f32, v_log_f32: y' = log2(1 + 2^x') - one plain and two
// quarter-rate issues where the natural-log form needs four and two (scale, |x|, max, fma).  Accuracy: for x' << 0 the sum 1 + 2^x' rounds exactly as
// 1 + e^-|x| does in the other form; for x' >> 0 the result carries v_log_f32's 1 ulp of ITS magnitude (relative 1e-7, the class of every fp32 operation
// here).  Range: 2^x' overflows at x' = 128, i.e. a pre-activation of 88.7 gives inf where F.softplus returns x (threshold 20) - far inside the range
// limit the fp16 planes already impose on this mode (65504), and loud.
// Round 6: rs = the inverse of the power of two the layer's weight planes carry (k_mlp_scales_h2): the accumulators hold x' / rs.  And the large-x branch of
// F.softplus (threshold 20: returns x): 2^x' overflows at x' = 128 (a pre-activation of 88.7) and round 5 returned inf there; now y' = med3(L, x', 128) with
// L = log2(1 + 2^x'): for x' < 128, x' <= L <= 128 and the median is L; beyond, L = inf and the median is x' - one v_med3_f32, no compare / select.
#ifndef HL_SP_NOCLAMP
#define HL_SP_NOCLAMP 0
#endif
template <int I0, int I1>
__device__ __forceinline__ void softplus_l2_r(f32x16 &v, float rs) {
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const float x = v[i] * rs, l = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(x));
        v[i] = HL_SP_NOCLAMP ? l : __builtin_amdgcn_fmed3f(l, x, 128.f);
    }
}
template <int I0, int I1>
__device__ __forceinline__ void softplus_b3_r(f32x16 &v) {   // registers I0 .. I1-1 of softplus16_b3, in place
    // (round 5: the packed-fp32 form - v_pk_mul / v_pk_add / v_pk_fma on pairs, |.| folded into v_exp_f32's source modifiers, 5 + 4 issues per pair instead
    //  of 8 + 4 - measured SLOWER on the same box, 27.6 against 27.0 ms per view: a packed fp32 instruction is two passes, and the pairs cost moves)
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(v[i]));
        v[i] = fmaf(0.693147180559945309f, __builtin_amdgcn_logf(1.f + e), b3_max0(v[i]));
    }
}
__device__ __forceinline__ f32x16 softplus16_b3(f32x16 v) {   // softplus_hidden with max(x, 0) as v_max_i32 (no canonicalising v_max in front)
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(v[i]));
        o[i] = fmaf(0.693147180559945309f, __builtin_amdgcn_logf(1.f + e), b3_max0(v[i]));
    }
    return o;
}
// NT output tiles x one 8-wide k-group: positions q0 .. q0+NT-1 of the chunk at `ch`; the six products, smallest first, two tiles at a time
// (24 fragment registers in flight; the partner wave covers the LDS latency)
template <int NT>
__device__ __forceinline__ void mma_b3(f32x16 (&acc)[NT], const u32x4 (&b)[3], const u32x4 *__restrict__ ch, int q0, int lane) {
    constexpr int PW[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += 2) {
        u32x4 w[2][3];
#pragma unroll
        for (int p = 2; p >= 0; --p)                           // in the order of first use (plane 2 of both tiles, plane 1, plane 0): the first MFMAs wait for two reads, not six
#pragma unroll
            for (int t = 0; t < 2; ++t) w[t][p] = ch[((q0 + t0 + t) * 3 + p) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[t][PW[i]]), __builtin_bit_cast(bf16x8, b[PB[i]]), acc[t0 + t], 0, 0, 0);
    }
}

// ---- fp16x2 (round 5): x = h0 + h1 with TWO fp16 planes (2 x 11 significand bits: |x - h0 - h1| <= 2^-20 |x| with the truncating split below,
// 2^-22 with nearest-even planes as the weights get at pack time), three partial products h0 w0 + h0 w1 + h1 w0 accumulated in fp32 on
// v_mfma_f32_32x32x16_f16 - HALF the MFMAs of the bf16x3 scheme at the same fp32-class error (the dropped h1 w1 is below 2^-20 |x w|): against
// the REFERENCE's golden renders rgb max-abs 1.2e-7 / 2.7e-7 / 9.5e-7 in a CPU emulation of exactly this arithmetic, 6e-8 / 1.5e-7 / 4.2e-7 for
// plain fp32 (profiles/r05_render_fp16x2.md).  h0 = the value with its low 13 mantissa bits cleared (v_and: what a round-toward-zero conversion
// to fp16 keeps), so the residual x - h0 is exact in fp32; both planes packed by v_cvt_pkrtz_f16_f32.  Range: |x| < 65504 (fp16); values below
// 2^-14 keep an ABSOLUTE error of 2^-24.
#ifndef HL_RENDER_SPLIT_RNE
// 1 (round 6, measured three times and NOT taken): nearest-even planes - h0 = the nearest fp16, h1 = the nearest fp16 of the residual (2^-24; a value beyond fp16
// becomes inf / NaN).  With hl_split2_rne's four-instruction form (hl_common.h) the kernels issue 7 % fewer vector instructions than with the truncating split below
// (7802 against 8385 in k_march_plw<2>) and every instruction involved issues at full rate (scripts/microbench/valu_rate.hip), yet a view takes 26.67 ms against
// 25.90 on the same box - also in the build where no kernel spills (the coarse kernel takes the compiler's five-instruction form, split_h2t<1>: with the asm form
// its sample loop needs 84 bytes of scratch).  The staging of these kernels is placed between the MFMAs by instruction class (sched_group_barrier), and an asm
// statement belongs to no class.
// 0: round 5's truncating split (h0 = the low 13 mantissa bits cleared, v_cvt_pkrtz: 2^-20, saturates silently - the clamped softplus and the sigma / rgb heads
// keep the renderer's activations far inside fp16's range).
#define HL_RENDER_SPLIT_RNE 0
#endif
template <int MODE = 0>   // 0: truncating; 1: nearest, left to the compiler (five instructions per pair); 2: nearest, hl_split2_rne's four - the same planes as 1, bit for bit
__device__ __forceinline__ void split_h2t(const f32x16 &v, int hi, u32x4 (&pl)[2]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x = v[8 * hi + 2 * q], y = v[8 * hi + 2 * q + 1];
        if constexpr (MODE == 2) {
            unsigned w0, w1;
            hl_split2_rne(x, y, w0, w1);
            pl[0][q] = w0; pl[1][q] = w1;
        } else if constexpr (MODE == 1) {
            unsigned w0, w1;
            hl_split2_rne_c(x, y, w0, w1);
            pl[0][q] = w0; pl[1][q] = w1;
        } else {
            const float hx = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffffe000u), hy = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, y) & 0xffffe000u);
            pl[0][q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(hx, hy));
            pl[1][q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x - hx, y - hy));
        }
    }
}
template <int MODE = 0> __device__ __forceinline__ void split_plt(const f32x16 &v, int hi, u32x4 (&pl)[3]) { split_b3t(v, hi, pl); }
template <int MODE = 0> __device__ __forceinline__ void split_plt(const f32x16 &v, int hi, u32x4 (&pl)[2]) { split_h2t<MODE>(v, hi, pl); }
// NT output tiles x one 8-wide k-group for NPL planes: the partial products, smallest first, two tiles at a time
template <int NT, int NPL>   // (both deduced from the arguments: the call sites sit inside macro arguments, where a template comma would split them)
__device__ __forceinline__ void mma_pl(f32x16 (&acc)[NT], const u32x4 (&b)[NPL], const u32x4 *__restrict__ ch, int q0, int lane) {
    if constexpr (NPL == 3) mma_b3<NT>(acc, b, ch, q0, lane);
    else {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        constexpr int PW[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += 2) {
            u32x4 w[2][2];
#pragma unroll
            for (int p = 1; p >= 0; --p)
#pragma unroll
                for (int t = 0; t < 2; ++t) w[t][p] = ch[((q0 + t0 + t) * 2 + p) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, w[t][PW[i]]), __builtin_bit_cast(h8, b[PB[i]]), acc[t0 + t], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ float wave_incl_scan_mul(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(v, d);
        if (lane >= d) v *= o;
    }
    return v;
}
__device__ __forceinline__ float wave_incl_scan_add(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}


// Phase B of the one-pass fine launch: the importance depths of ONE 32-ray tile, drawn by the wave that owns it - k_importance's arithmetic (up_sample / sample_pdf,
// renderer.py:158-170, 533-563), one ray after the other with wave-wide scans, the cdf search and a bitonic sort of the new depths in the wave's own LDS (s_w: 128
// floats, s_z: 256; LDS operations of one wave execute in order, so no barrier is needed where the workgroup version has them).  N coarse = N new <= 128; the coarse
// depths are linspace(near, far).  zn_tile = the tile's [N][32] block of new depths, sorted per ray; zbuf = 16 N floats of LDS (the stage of a half tile).
__device__ __forceinline__ void importance_tile(const float4 *__restrict__ vc_tile, const float *__restrict__ u, const float *__restrict__ rays_d, const float *__restrict__ near,
                                                const float *__restrict__ far, long long R, int N, long long tile, int lane, float *s_w, float *s_z, float *zbuf,
                                                float *zn_tile) {
    const float *sig = reinterpret_cast<const float *>(vc_tile);      // .x of the record of (sample i, ray j) at (32 i + j) * 4
    for (int r8 = 0; r8 < 32; r8 += 8) {
        float sg[8][2];                                               // the densities of "this lane's" two samples of the next eight rays: one 128-byte line per sample
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int i = 64 * b + lane;
                sg[rr][b] = i < N ? sig[(32LL * i + r8 + rr) * 4] : 0.f;
            }
        // ... and everything else the eight rays read from memory, requested up front as well (near, far, direction, this lane's two uniforms): walking the rays one
        // after the other with the loads where they are used exposes a round trip to HBM per ray and stage - 1.1 ms per 512x512 view for this phase, half of it waiting
        float nr8[8], fr8[8], dx8[8], dy8[8], dz8[8], ua8[8], ub8[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const long long rw = tile * 32 + r8 + rr, ry = rw < R ? rw : R - 1;
            nr8[rr] = near[ry]; fr8[rr] = far[ry];
            dx8[rr] = rays_d[ry * 3]; dy8[rr] = rays_d[ry * 3 + 1]; dz8[rr] = rays_d[ry * 3 + 2];
            ua8[rr] = lane < N ? u[ry * N + lane] : 0.f;
            ub8[rr] = lane + 64 < N ? u[ry * N + lane + 64] : 0.f;
        }
        for (int rr = 0; rr < 8; ++rr) {
            const int j = r8 + rr;
            const float nr = nr8[0], fr = fr8[0];
            const float dxx = dx8[0], dyy = dy8[0], dzz = dz8[0];
            const float dn = sqrtf(dxx * dxx + dyy * dyy + dzz * dzz);
            auto zval = [&](int i) -> float {
                const float t = linspace01(i, N);
                return nr * (1.f - t) + fr * t;
            };
            // weights w_i = alpha_i * prod_{k<i}(1 - alpha_k + 1e-10)
            float carry = 1.f;
            for (int base = 0; base < N; base += 64) {
                const int i = base + lane;
                float alpha = 0.f, zi = 0.f;
                if (i < N) {
                    zi = zval(i);
                    float dist = (i + 1 < N) ? zval(i + 1) - zi : 1e10f;
                    dist = dist * dn;
                    const float sraw = base == 0 ? sg[0][0] : sg[0][1];
                    alpha = 1.f - expf(-softplus_exact(sraw) * dist);
                    s_z[i] = zi;
                }
                const float fct = (i < N) ? (1.f - alpha + 1e-10f) : 1.f;
                const float incl = wave_incl_scan_mul(fct, lane);
                float excl = __shfl_up(incl, 1);
                if (lane == 0) excl = 1.f;
                if (i < N) s_w[i] = alpha * (carry * excl);
                carry *= __shfl(incl, 63);
            }
            __builtin_amdgcn_wave_barrier();
            // pdf over w[1..N-2] (+1e-5), cdf[0]=0, cdf[m]=sum_{i<=m} pdf_i   (N-1 entries)
            const int M = N - 2;
            float tot = 0.f;
            for (int i = 1 + lane; i <= M; i += 64) tot += s_w[i] + 1e-5f;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d);
            float run = 0.f;
            for (int base = 1; base <= M; base += 64) {
                const int i = base + lane;
                const float pr = (i <= M) ? (s_w[i] + 1e-5f) / tot : 0.f;
                const float incl = wave_incl_scan_add(pr, lane) + run;
                __builtin_amdgcn_wave_barrier();
                if (i <= M) s_w[i] = incl;
                run = __shfl(incl, 63);
            }
            if (lane == 0) s_w[0] = 0.f;
            __builtin_amdgcn_wave_barrier();
            // inverse CDF: idx = #(cdf <= u) (searchsorted right=True) over cdf[0..M]; this lane's two uniforms (q = lane, lane + 64) searched together: two
            // independent chains of LDS reads instead of one after the other
            const int nc = M + 1;
            const int q0 = lane, q1 = lane + 64;
            const float u0 = ua8[0], u1 = ub8[0];
            int lo0 = 0, hi0 = nc, lo1 = 0, hi1 = nc;
            while (lo0 < hi0 || lo1 < hi1) {
                const int m0 = (lo0 + hi0) >> 1, m1 = (lo1 + hi1) >> 1;
                const float w0 = s_w[m0 < nc ? m0 : nc - 1], w1 = s_w[m1 < nc ? m1 : nc - 1];
                if (lo0 < hi0) { if (w0 <= u0) lo0 = m0 + 1; else hi0 = m0; }
                if (lo1 < hi1) { if (w1 <= u1) lo1 = m1 + 1; else hi1 = m1; }
            }
            auto inv_cdf = [&](int lo, float uq) -> float {
                const int below = max(lo - 1, 0), above = min(lo, nc - 1);
                const float c0 = s_w[below], c1 = s_w[above];
                const float b0 = 0.5f * (s_z[below + 1] + s_z[below]);
                const float b1 = 0.5f * (s_z[above + 1] + s_z[above]);
                float den = c1 - c0;
                den = den < 1e-5f ? 1.f : den;
                const float t = (uq - c0) / den;
                return b0 + t * (b1 - b0);
            };
            const float inf_ = __builtin_inff();
            const float z0 = q0 < N ? inv_cdf(lo0, u0) : inf_, z1 = q1 < N ? inv_cdf(lo1, u1) : inf_;
            // Sorted by RANK instead of a bitonic network in LDS (28 dependent LDS round trips per ray): position of z_q = #{z_j < z_q} + #{j < q : z_j = z_q}, the other
            // lanes' values broadcast through v_readlane - no memory, ~600 plain instructions.  The sorted list is the same (the coarse depths are sorted already; the
            // compositing merges), and equal depths are interchangeable.
            // (keys (depth bits, index): depths are >= 0, so their bit patterns order like the values, and the index breaks ties - one 64-bit compare per pair)
            const unsigned long long k0 = ((unsigned long long)__builtin_bit_cast(unsigned, z0) << 32) | (unsigned)q0;
            const unsigned long long k1 = ((unsigned long long)__builtin_bit_cast(unsigned, z1) << 32) | (unsigned)q1;
            int r0 = 0, r1 = 0;
#pragma unroll 4
            for (int jl = 0; jl < 64; ++jl) {      // (fully unrolled, the 128 broadcast values are all kept in scalar registers at once and spill by the hundred)
                const unsigned long long a0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, z0), jl) << 32) | (unsigned)jl;          // key of q = jl
                const unsigned long long a1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, z1), jl) << 32) | (unsigned)(jl + 64);   // key of q = 64 + jl
                r0 += (a0 < k0 ? 1 : 0) + (a1 < k0 ? 1 : 0);
                r1 += (a0 < k1 ? 1 : 0) + (a1 < k1 ? 1 : 0);
            }
            // (by rank into the half-tile stage [sample][16 rays]: written out below as 64-byte row pieces - 4-byte stores straight into the tile-major rows cost
            //  32 bytes of HBM write each: 1.1 GB per 512x512 view for 134 MB of depths)
            if (q0 < N) zbuf[r0 * 16 + (j & 15)] = z0;
            if (q1 < N) zbuf[r1 * 16 + (j & 15)] = z1;
            if ((j & 15) == 15) {
                __builtin_amdgcn_wave_barrier();
                for (int e = lane * 4; e < N * 16; e += 256) {
                    const int i = e >> 4, c = e & 15;
                    *reinterpret_cast<f32x4 *>(zn_tile + 32LL * i + (j & 16) + c) = *reinterpret_cast<const f32x4 *>(zbuf + e);
                }
                __builtin_amdgcn_wave_barrier();
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 7; ++k) {
#pragma unroll
                for (int b = 0; b < 2; ++b) sg[k][b] = sg[k + 1][b];
                nr8[k] = nr8[k + 1]; fr8[k] = fr8[k + 1]; dx8[k] = dx8[k + 1]; dy8[k] = dy8[k + 1]; dz8[k] = dz8[k + 1]; ua8[k] = ua8[k + 1]; ub8[k] = ub8[k + 1];
            }
        }
    }
}

#ifndef HL_H2_K
#define HL_H2_K 4   // VALU instructions asked for behind every MFMA of a hidden-layer chunk in the fp16x2 kernel (12 MFMAs, ~48 VALU of preparation with the
                    // log2-domain softplus; same box, ms per 512x512 view: 3: 25.56, 4: 25.74, 5: 25.81 - 26.00, 6: 26.35; natural-log softplus at 5: 26.71)
#endif
// ACTS (training; round 5): the evaluate pass of the fitting step on this kernel - every activation of the MLP is also written to the
// activation matrix (row = unit, column = sample point; the rows k_mlp_bwd and k_wgrad read), a workgroup takes the sample range blockIdx.y * s_per ... of its 256 rays
// (a fitting batch has few rays), one workgroup per CU (the stores need registers).  Softplus outputs are stored in natural units (x ln 2 in the log2 domain).
// FUSE (round 6): the ONE-PASS fine launch.  Phase B - before anything of the MLP is set up, every wave draws the importance depths of its 32 rays from the coarse
// launch's records (importance_tile: its LDS scratch is the wave's vinit block, not yet in use) and writes them, sorted, to a tile-major scratch only it reads back.
// Phase C - the sample loop evaluates those depths and, instead of storing records, composites: every lane keeps its ray's transmittance / sums and a cursor into
// the coarse records, and after each new sample emits the coarse samples in front of it (one 16-byte record load each, the next one requested as the cursor
// moves) and then the sample itself - k_composite's merge order and arithmetic, sample by sample, so the image equals the four-launch pipeline's bit for bit.
// Per 512x512 view: no k_importance / k_composite launches, no 537 MB of fine records written and read, no second read of the coarse records by a merge kernel.
template <int NPL, bool ACTS = false, bool FUSE = false>
__global__ __launch_bounds__(512, ACTS ? 1 : 2) void k_march_plw(const MarchArgs a, const unsigned short *__restrict__ packed_b3) {
    static_assert(!(ACTS && FUSE), "the training pass stores records");
    constexpr int SPL = !HL_RENDER_SPLIT_RNE ? 0 : ((ACTS || FUSE) ? 2 : 1);   // (the coarse kernel's register allocation spills with the asm form; the planes are the same)
    constexpr int B3R_SLOT_U4 = PLW_SLOT_U4<NPL>, B3_CH_U4 = PLW_CH_U4<NPL>;      // (shadow the bf16x3 constants of k_march_b3)
    constexpr size_t B3_BYTES = PLW_BYTES<NPL>;
    constexpr int NMF = NPL == 3 ? 24 : 12;                                        // MFMAs of a chunk (4 tiles x 6 | 3 products)
#ifndef HL_H2_NO_LOG2
    constexpr bool LOG2D = NPL == 2;                                               // softplus in the log2 domain (softplus_l2_r; scaled planes and tables)
#else
    constexpr bool LOG2D = false;
#endif
    constexpr float L2E = 1.44269504088896341f, LN2 = 0.693147180559945309f;
    extern __shared__ __attribute__((aligned(16))) float ldsb[];   // [ring: 2 x 24 KB][small 4 KB][per wave: the 2 x 16 x 64 accumulator image of views_linear's bias + direction part, 8 KB]
    constexpr int NT = 512, NST = B3R_SLOT_U4 / NT;   // threads; u32x4 per thread and chunk pair (3 | 2)
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const long long wg = (long long)xcd * q8 + (xcd < r8 ? xcd : r8) + (blockIdx.x >> 3);
    const long long tile = wg * 8 + (tid >> 6);
    const long long ray = tile * 32 + (lane & 31);
    const bool valid = ray < a.R;
    const long long rc = valid ? ray : a.R - 1;
    const long long tiles_n = (a.R + 31) / 32;
    const long long zt_base = (tile < tiles_n ? tile : tiles_n - 1) * 32 * (long long)a.S + (lane & 31);

    u32x4 *ring = reinterpret_cast<u32x4 *>(ldsb);
    float *small = ldsb + 2 * B3R_SLOT_U4 * 4;
    f32x4 *vinit = reinterpret_cast<f32x4 *>(small + SMALL_FLOATS) + (tid >> 6) * 512;
    // the layers' plane scales and their inverses (k_mlp_scales_h2; 1 outside the log2-domain fp16x2 mode, whose planes are the only scaled ones)
    const float *gsc = a.packed + NCH_FULL * CHUNK_FLOATS + SM_SC;
    const float sc0 = LOG2D ? gsc[0] : 1.f, sc1 = LOG2D ? gsc[1] : 1.f, sc2 = LOG2D ? gsc[2] : 1.f, scF = LOG2D ? gsc[3] : 1.f, scV = LOG2D ? gsc[4] : 1.f;
    const float rs0 = LOG2D ? gsc[8] : 1.f, rs1 = LOG2D ? gsc[9] : 1.f, rs2 = LOG2D ? gsc[10] : 1.f, rsF = LOG2D ? gsc[11] : 1.f, rsV = LOG2D ? gsc[12] : 1.f;
    for (int i = tid; i < SMALL_FLOATS; i += NT) {
        float v = a.packed[NCH_FULL * CHUNK_FLOATS + i];
        if constexpr (LOG2D) {   // biases in front of a softplus x log2(e), every bias x its layer's plane scale (it initialises the scaled accumulators); head weights behind a softplus x ln 2
            v *= (i < SM_BF || (i >= SM_BV && i < SM_AW)) ? L2E : ((i >= SM_AW && i < SM_AB) ? LN2 : 1.f);
            v *= i < SM_B1 ? sc0 : (i < SM_B2 ? sc1 : (i < SM_BF ? sc2 : (i < SM_BV ? scF : (i < SM_AW ? scV : 1.f))));
        }
        small[i] = v;
    }
    const u32x4 *gb3 = reinterpret_cast<const u32x4 *>(packed_b3);
    if constexpr (FUSE) {   // phase B: this wave's importance depths
        if (tile < tiles_n && !(a.flags & 0x10000u)) {             // (0x10000: developer timing switch - the depths of an earlier call are still in the scratch)
            // (LDS: the weights / cdf and the depths of the ray in the wave's share of the weight ring, not yet loaded; the half-tile stage of the sorted depths -
            //  8 KB - in the wave's vinit block, not yet written)
            float *s_w = reinterpret_cast<float *>(ring) + (tid >> 6) * 384, *s_z = s_w + 128;
            importance_tile(a.fz_vc + tile * 32 * (long long)a.fz_N, a.fz_u, a.rays_d, a.near, a.far, a.R, a.fz_N, tile, lane, s_w, s_z,
                            reinterpret_cast<float *>(vinit), a.fz_zn + tile * 32 * (long long)a.S);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // the depths are read back below through a.z (= a.fz_zn): stores complete, then no stale line
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();                                        // (the ring is loaded next: every wave is done with its scratch in it)
    }
#pragma unroll
    for (int q = 0; q < 2 * NST; ++q) ring[q * NT + tid] = gb3[q * NT + tid];                  // chunk pairs 0, 1 -> slots 0, 1
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)packed_b3, (short)0, (int)B3_BYTES, 0x00020000);
    const int wv = tid * 16;
    auto ldw = [&](int u4_index) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(rsw, wv, u4_index * 16, 0); };
    u32x4 st[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) st[q] = ldw(2 * B3R_SLOT_U4 + q * NT);                     // pair 2 staged
    int cur = 0;

    const float ox = a.rays_o[rc * 3 + 0], oy = a.rays_o[rc * 3 + 1], oz = a.rays_o[rc * 3 + 2];
    const float dx = a.rays_d[rc * 3 + 0], dy = a.rays_d[rc * 3 + 1], dz = a.rays_d[rc * 3 + 2];
    const float nr = a.near[rc], fr_ = a.far[rc];
    const int S = a.S;
    const float offH = (float)(1.0 / (double)a.H);
    const float bmin0 = a.bounds[0], bmin1 = a.bounds[1], bmin2 = a.bounds[2];
    const float bext0 = a.bounds[3] - bmin0, bext1 = a.bounds[4] - bmin1, bext2 = a.bounds[5] - bmin2;

    // view-direction encoding, this half's 14 of the 27 (+1 pad) entries (as k_march), split once per ray
    {
        u32x4 bev0[NPL], bev1[NPL];
        f32x16 ev;
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        const float vd[3] = {dx / nrm, dy / nrm, dz / nrm};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float val = 0.f;
            if (s < 14) {
                const int kl = s, kh = s + 14;
                const int jl = (kl - 3) / 3, cl = (kl - 3) % 3, jh = (kh - 3) / 3, ch = (kh - 3) % 3;
                const float argl = kl < 3 ? 0.f : ((jl & 1) ? 1.57079632679489661923f : 0.f) + vd[kl < 3 ? 0 : cl] * (float)(1 << (jl >> 1));
                const float argh = ((jh & 1) ? 1.57079632679489661923f : 0.f) + vd[ch] * (float)(1 << (jh >> 1));
                if (kl < 3) {
                    const float sh = sinf(argh);
                    val = half ? sh : vd[kl];
                } else if (kh >= 27) {
                    const float sl = sinf(argl);
                    val = half ? 0.f : sl;
                } else {
                    val = sinf(half ? argh : argl);
                }
            }
            ev[s] = val;
        }
        split_plt<SPL>(ev, 0, bev0);
        split_plt<SPL>(ev, 1, bev1);
        if constexpr (ACTS) {   // the encoding is a row block of the activation matrix, the same for every sample of the ray: written here for the whole sample range
            const int s_lo_ = (int)blockIdx.y * a.s_per, s_hi_ = min(a.S, s_lo_ + a.s_per);
            const unsigned as4 = (unsigned)a.act_stride * 4u;
            const i32x4 rs_ = matrix_rsrc(a.act, (unsigned)ACT_ROWS * as4);
            if (tile * 32 < a.R)
                for (int s = s_lo_; s < s_hi_; ++s) {
                    const unsigned col4 = (unsigned)(a.act_off + zt_base + 32LL * s) * 4u;
#pragma unroll
                    for (int j = 0; j < 14; ++j)
                        if (j + 14 * half < 27) hidden_store(rs_, col4 + (unsigned)half * 14u * as4, (unsigned)(ROW_EV + j) * as4, ev[j]);
                }
        }
    // views_linear's direction part does not depend on the sample: bias + W_dir enc(dir) is formed once per ray (chunk 32, fragments straight
    // from global memory) and parked in LDS as the accumulator image every sample starts views_linear from
        f32x16 V0[2];
        load_bias_global<2>(V0, a.packed + NCH_FULL * CHUNK_FLOATS + SM_BV, half);
        if constexpr (LOG2D) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) V0[t][r] *= L2E * scV;
        }
        mma_pl(V0, bev0, gb3 + (B3_NCH - 1) * B3_CH_U4, 0, lane);
        mma_pl(V0, bev1, gb3 + (B3_NCH - 1) * B3_CH_U4, 2, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) vinit[(t * 4 + q) * 64 + lane] = f32x4{V0[t][4 * q], V0[t][4 * q + 1], V0[t][4 * q + 2], V0[t][4 * q + 3]};
    }

    int s_lo = 0, s_hi = S;
    if constexpr (ACTS) {
        s_lo = (int)blockIdx.y * a.s_per;
        s_hi = min(S, s_lo + a.s_per);
    }
    const unsigned act_stride4 = ACTS ? (unsigned)a.act_stride * 4u : 0u;
    const i32x4 act_rs = matrix_rsrc(a.act, ACTS ? (unsigned)ACT_ROWS * act_stride4 : 0u);
    float zc;
    if constexpr (FUSE) zc = a.z[zt_base + 32LL * s_lo];                                    // (the one-pass launch reads its own tile-major depths: one address form to keep)
    else if (a.z) zc = a.z_tiled ? a.z[zt_base + 32LL * s_lo] : a.z[rc * S + s_lo];
    else zc = nr * (1.f - linspace01(s_lo, S)) + fr_ * linspace01(s_lo, S);
    __syncthreads();

    // Ring of two 24 KB slots, each a PAIR of chunks (8 fragment positions x 3 planes).  While pair P is multiplied: slot `cur` holds P, the other
    // slot holds (or is being filled with) P+1, the staging registers hold (or are receiving) P+2.  B3_ADV(g) in front of chunk g: nothing for
    // the second chunk of a pair; for the first: barrier (everybody is done with pair P-1, the writes of P are visible), flip, write the staged
    // pair P+1 into the slot P-1 released, start loading pair P+2.  One barrier per 48 MFMAs.
#define B3_ADV(g)                                                                                                            \
    if (((g) & 1) == 0) {                                                                                                    \
        __syncthreads();                                                                                                     \
        cur ^= B3R_SLOT_U4;                                                                                                  \
        _Pragma("unroll") for (int q_ = 0; q_ < NST; ++q_) ring[(cur ^ B3R_SLOT_U4) + q_ * NT + tid] = st[q_];               \
        _Pragma("unroll") for (int q_ = 0; q_ < NST; ++q_) st[q_] = ldw(((((g) >> 1) + 2) % B3W_NPAIR) * B3R_SLOT_U4 + q_ * NT); \
    }
#define B3_AT(g) (ring + cur + ((g) & 1) * B3_CH_U4)
#define SP_R(I0_, I1_, V_, RS_) { if constexpr (LOG2D) softplus_l2_r<I0_, I1_>(V_, RS_); else softplus_b3_r<I0_, I1_>(V_); }
    // The two waves of a SIMD (w and w + 4) run the same chunks between the same barriers; left alone they prepare operands (VALU) at the same
    // time and multiply (MFMA) at the same time and the two pipes take turns.  B3_VM orders every chunk as [prepare the next operand, multiply]
    // in waves 0-3 and as [multiply, prepare] in waves 4-7 (ROT): one wave's VALU phase meets the other's MFMA phase.
#define B3_VM(V_, M_)                                                                      \
    if constexpr (ROT) { M_; __builtin_amdgcn_sched_barrier(0); V_; __builtin_amdgcn_sched_barrier(0); } \
    else { V_; __builtin_amdgcn_sched_barrier(0); M_; __builtin_amdgcn_sched_barrier(0); }
    // B3_MV24, the chunks of the hidden layers: ONE scheduling region with the chunk's 24 MFMAs and the preparation of the next operand - softplus of
    // one HALF of a tile + its exact three-way split, ~100 VALU instructions - asked of the scheduler as [1 MFMA, K_ VALU] x 24 (sched_group_barrier):
    // the preparation rides in the shadow of the matrix pipe instead of standing as a block beside the chunk.  The softplus of a tile is spread
    // over both of its chunks (4.2 VALU per MFMA each; the whole tile in every second chunk, 6.5 against 1.8, gains half as much).
    // Same-box A/B, ms per 512x512 view: blocks 43.5 - 44.0, whole-tile softplus 42.4, this 41.7 - 42.0 (profiles/r04_render_b3_ablations.md).
#define B3_MV24(K_, M_, ...)                                                               \
    { __VA_ARGS__; M_;                                                                     \
      _Pragma("unroll") for (int g_ = 0; g_ < NMF; ++g_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NPL == 3 ? K_ : HL_H2_K, 0); } \
      __builtin_amdgcn_sched_barrier(0); }

    // FUSE: streaming merge + compositing state of this lane's ray (k_composite's arithmetic in k_composite's order; both lane halves of a ray carry the shared
    // part, half 0 the red sum, half 1 green and blue): eleven registers per lane across the sample's MLP (252 of 256 in use, no spill - the addresses below are
    // formed where they are used for that reason).  The record at the cursor is requested while views_linear runs and is in flight until the emission.
    // registers per lane: transmittance; three sums - lane half 0 (red, sum w, sum w z), half 1 (green, blue, -); the pending sample (its alpha needs the NEXT
    // depth): depth, raw density, two raw colours (half 0: red, -; half 1: green, blue); the cursor
    float cT = 1.f, cA = 0.f, cB = 0.f, cC = 0.f;
    float pzd = 0.f, ps = 0.f, pc0 = 0.f, pc1 = 0.f;
    bool has_p = false;
    int ci = 0;                                                 // cursor into the coarse samples
    const int Nc = FUSE ? a.fz_N : 0;
    // (the addresses are formed where they are used, from a lane index the optimiser cannot see through: hoisted out of the sample loop they would be live across
    //  the MLP and spill)
    auto opaque_tid = [&]() __attribute__((always_inline)) -> int { int t_ = tid; asm volatile("" : "+v"(t_)); return t_; };
    auto vcp = [&]() __attribute__((always_inline)) -> const float2 * {                        // this lane half's two entries of the record of coarse sample 0: (sigma, r) | (g, b)
        const int t_ = opaque_tid();
        const long long tl = wg * 8 + (t_ >> 6);
        return reinterpret_cast<const float2 *>(a.fz_vc + (tl < tiles_n ? tl : tiles_n - 1) * 32 * (long long)Nc + (t_ & 31)) + ((t_ >> 5) & 1);
    };
    auto zco = [&](int i) __attribute__((always_inline)) -> float {
        if (i >= Nc) return __builtin_inff();
        const float t = linspace01(i, Nc);
        return nr * (1.f - t) + fr_ * t;
    };
    float2 crec = make_float2(0.f, 0.f), crec1 = make_float2(0.f, 0.f);   // this half's entries of the records at the cursor and behind it
    auto lower_half = [&](float v) __attribute__((always_inline)) -> float {                    // the value lane (lane & 31) holds: v_permlane32_swap, no LDS (both halves of a ray are always active together)
        const unsigned x_ = __builtin_bit_cast(unsigned, v);
        return __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(x_, x_, false, false)[0]);
    };
    auto finish_pending = [&](float dist) __attribute__((always_inline)) {
        const float alpha = dist == 1e10f ? comp_alpha_last(ps) : comp_alpha(ps, dist);
        const float w = alpha * cT;
        cA += comp_sigmoid(pc0) * w;
        cB += (half ? comp_sigmoid(pc1) : 1.f) * w;              // (half 0: 1 w = w exactly - the sum of the weights)
        cC += w * pzd;
        cT *= (1.f - alpha + 1e-7f);
    };
    auto emit = [&](float z, float sg_, float c0, float c1) __attribute__((always_inline)) {
        if (has_p) finish_pending(z - pzd);
        pzd = z; ps = sg_; pc0 = c0; pc1 = c1; has_p = true;
    };
    // Every coarse sample at or in front of zlim.  The record at the cursor was requested while views_linear ran; inside a run every further record costs a round trip
    // to L2 (measured alternatives: two / four records requested together need 4 / 8 more registers from the views stage on and the allocator starts to spill -
    // 6 ... 25 registers of scratch - which costs more than the round trips: every reload is a vmcnt(0) in front of the weight ring's counted waits).
#ifndef HL_FUSE_LQ
#define HL_FUSE_LQ 0
#endif
#if HL_FUSE_LQ
    // A / B variant: the LQ records at the cursor ... cursor + LQ - 1 of every lane gathered into a per-wave LDS block [LQ][64 lanes] x 16 bytes by LDS-DMA while
    // views_linear runs (no registers); waited for in front of the ring advance (where the only other loads in flight are the ring's staged ones, needed there anyway)
    constexpr int LQ = 4;
    auto lqp = [&]() __attribute__((always_inline)) -> f32x4 * { const int t_ = opaque_tid(); return reinterpret_cast<f32x4 *>(small + SMALL_FLOATS) + 8 * 512 + (t_ >> 6) * (LQ * 64); };
    auto lq_request = [&]() __attribute__((always_inline)) {
        const int t_ = opaque_tid();
        const long long tl = wg * 8 + (t_ >> 6);
        const float4 *base_ = a.fz_vc + (tl < tiles_n ? tl : tiles_n - 1) * 32 * (long long)Nc;
        const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc((void *)base_, (short)0, Nc * 512, 0x00020000);
        f32x4 *q_ = lqp();
#pragma unroll
        for (int k = 0; k < LQ; ++k) {
            const int row = ci + k < Nc ? ci + k : Nc - 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (__attribute__((address_space(3))) void *)(q_ + k * 64), 16, (row * 32 + (t_ & 31)) * 16, 0, 0, 0);
        }
    };
    auto emit_coarse_until = [&](float zlim) __attribute__((always_inline)) {
        const f32x4 *q_ = lqp() + (opaque_tid() & 63);
        int k = 0;
        while (zco(ci) <= zlim) {
            f32x4 rec;
            if (k < LQ) rec = q_[k * 64];
            else { const float4 r_ = a.fz_vc[((wg * 8 + (opaque_tid() >> 6)) < tiles_n ? (wg * 8 + (opaque_tid() >> 6)) : tiles_n - 1) * 32 * (long long)Nc + 32LL * ci + (opaque_tid() & 31)]; rec = f32x4{r_.x, r_.y, r_.z, r_.w}; }
            emit(zco(ci), rec[0], half ? rec[2] : rec[1], rec[3]);
            ++ci; ++k;
        }
    };
#else
    auto emit_coarse_until = [&](float zlim) __attribute__((always_inline)) {
        while (zco(ci) <= zlim) {
            emit(zco(ci), lower_half(crec.x), half ? crec.x : crec.y, crec.y);      // (the density sits in half 0)
            ++ci;
            crec = crec1;                                                           // (requested one step earlier: two records are in flight inside a run)
            if (ci + 1 < Nc) crec1 = vcp()[64LL * (ci + 1)];
        }
    };
#endif
    auto body = [&](auto rotc) {
    constexpr bool ROT = decltype(rotc)::value;
    for (int s = s_lo; s < s_hi; ++s) {
        float zn = 0.f;
        if (s + 1 < S) {
            if constexpr (FUSE) {   // (address formed here from an opaque lane index: hoisted out of the loop it is spilled, and its reload costs a vmcnt(0) per sample)
                int t_ = tid;
                asm volatile("" : "+v"(t_));
                const long long tl = wg * 8 + (t_ >> 6);
                zn = a.z[(tl < tiles_n ? tl : tiles_n - 1) * 32 * (long long)a.S + (t_ & 31) + 32LL * (s + 1)];
            }
            else if (a.z) zn = a.z_tiled ? a.z[zt_base + 32LL * (s + 1)] : a.z[rc * S + s + 1];
            else { const float t = linspace01(s + 1, S); zn = nr * (1.f - t) + fr_ * t; }
        }
        // ---- tri-plane features of this half (fp32, exactly as k_march)  [renderer.py:502-531] ----
        const float px = ox + dx * zc, py = oy + dy * zc, pz = oz + dz * zc;
        const float nx = 2.f * (px - bmin0) / bext0 - 1.f;
        const float ny = 2.f * (py - bmin1) / bext1 - 1.f;
        const float nz = 2.f * (pz - bmin2) / bext2 - 1.f;
        f32x16 f;
        f[15] = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int qlo = i, qhi = (i + 5 > 8) ? 8 : i + 5;
            const int q = half ? qhi : qlo;
            const int p = half ? qhi / 3 : qlo / 3, g = half ? qhi % 3 : qlo % 3;
            float gu = (p == 2) ? nz : nx;
            float gv = (p == 1) ? nz : ny;
            gu = (g == 1) ? gu + offH : gu;
            gv = (g == 2) ? gv + offH : gv;
            const float ix = ((gu + 1.f) * (float)a.W - 1.f) / 2.f;
            const float iy = ((gv + 1.f) * (float)a.H - 1.f) / 2.f;
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float x1f = x0f + 1.f, y1f = y0f + 1.f;
            float w_nw = (x1f - ix) * (y1f - iy), w_ne = (ix - x0f) * (y1f - iy);
            float w_sw = (x1f - ix) * (iy - y0f), w_se = (ix - x0f) * (iy - y0f);
            const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
            const bool vx0 = (x0 >= 0) & (x0 < a.W), vx1 = (x1 >= 0) & (x1 < a.W);
            const bool vy0 = (y0 >= 0) & (y0 < a.H), vy1 = (y1 >= 0) & (y1 < a.H);
            w_nw = (vx0 & vy0) ? w_nw : 0.f;
            w_ne = (vx1 & vy0) ? w_ne : 0.f;
            w_sw = (vx0 & vy1) ? w_sw : 0.f;
            w_se = (vx1 & vy1) ? w_se : 0.f;
            const int cx0 = min(max(x0, 0), a.W - 1), cx1 = min(max(x1, 0), a.W - 1);
            const int cy0 = min(max(y0, 0), a.H - 1), cy1 = min(max(y1, 0), a.H - 1);
            const float4 *pl = a.planes + (long long)q * a.H * a.W;
            const float4 t_nw = pl[cy0 * a.W + cx0], t_ne = pl[cy0 * a.W + cx1];
            const float4 t_sw = pl[cy1 * a.W + cx0], t_se = pl[cy1 * a.W + cx1];
            const bool live = half ? (i + 5 <= 8) : true;
            const float r0 = t_nw.x * w_nw + t_ne.x * w_ne + t_sw.x * w_sw + t_se.x * w_se;
            const float r1 = t_nw.y * w_nw + t_ne.y * w_ne + t_sw.y * w_sw + t_se.y * w_se;
            const float r2 = t_nw.z * w_nw + t_ne.z * w_ne + t_sw.z * w_sw + t_se.z * w_se;
            f[3 * i + 0] = live ? r0 : 0.f;
            f[3 * i + 1] = live ? r1 : 0.f;
            f[3 * i + 2] = live ? r2 : 0.f;
        }
        const bool act_on = ACTS && tile * 32 < a.R;
        unsigned acol = 0;       // byte offset of this sample point's column (+ this half's 4 rows) in the activation matrix
        auto act_rows = [&](int row0, auto &h, float sc) {          // rows row0 + unit_of(t, r, half) of this column (sc: log2 units -> natural)
            if constexpr (ACTS) {
                if (act_on) {
                    constexpr int NT_ = sizeof(h) / sizeof(h[0]);
#pragma unroll
                    for (int t = 0; t < NT_; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) hidden_store(act_rs, acol, (unsigned)(row0 + unit_of(t, r, 0)) * act_stride4, h[t][r] * sc);
                }
            }
        };
        if constexpr (ACTS) {
            const unsigned col4 = (unsigned)(a.act_off + zt_base + 32LL * s) * 4u;
            acol = col4 + (unsigned)half * 4u * act_stride4;
            if (act_on) {
#pragma unroll
                for (int j = 0; j < 15; ++j)
                    if (j + 15 * half < 27) hidden_store(act_rs, col4 + (unsigned)half * 15u * act_stride4, (unsigned)(ROW_F + j) * act_stride4, f[j]);
            }
        }
        constexpr float SPU = LOG2D ? LN2 : 1.f;                    // softplus outputs of this kernel -> natural units
        u32x4 bf0[NPL], bf1[NPL], ba[NPL], bb[NPL];
        split_plt<SPL>(f, 0, bf0);
        split_plt<SPL>(f, 1, bf1);
        // ---- MLP  [renderer.py:134-156]: chunk g = fragment positions 4g .. 4g+3.  Software-pipelined: the operand of chunk g+1 is prepared
        // (softplus of a tile at its first use, three-way split of one half) in the same scheduling region as the MFMAs of chunk g, and the
        // (__builtin_amdgcn_sched_group_barrier patterns over regions of this size do not finish compiling) ----
        f32x16 X[4], Y[4];
        load_bias<4>(X, small + SM_B0, half);
        mma_pl(X, bf0, B3_AT(0), 0, lane);                                       // L0: chunks 0, 1
        B3_ADV(1) mma_pl(X, bf1, B3_AT(1), 0, lane);
        load_bias<4>(Y, small + SM_B1, half);
        SP_R(0, 8, X[0], rs0);                                             // (the second half: behind the first chunk of the layer)
        split_plt<SPL>(X[0], 0, ba);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                               // L1: chunks 2..9 = (tile k of X, half 0 | 1)
            B3_ADV(2 + 2 * k)
            B3_MV24(5, mma_pl(Y, ba, B3_AT(2 + 2 * k), 0, lane), { SP_R(8, 16, X[k], rs0); split_plt<SPL>(X[k], 1, bb); })
            B3_ADV(3 + 2 * k)
            B3_MV24(5, mma_pl(Y, bb, B3_AT(3 + 2 * k), 0, lane), if (k < 3) { SP_R(0, 8, X[k + 1 < 4 ? k + 1 : 3], rs0); split_plt<SPL>(X[k + 1 < 4 ? k + 1 : 3], 0, ba); })
        }
        act_rows(ROW_X0, X, SPU);
        load_bias<4>(X, small + SM_B2, half);
        B3_ADV(10) mma_pl(X, bf0, B3_AT(10), 0, lane);                          // L2 (features): chunks 10, 11; the first hidden operand rides along
        B3_ADV(11)
        B3_VM({ SP_R(0, 8, Y[0], rs1); split_plt<SPL>(Y[0], 0, ba); }, mma_pl(X, bf1, B3_AT(11), 0, lane))
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                               // L2 (hidden): chunks 12..19
            B3_ADV(12 + 2 * k)
            B3_MV24(5, mma_pl(X, ba, B3_AT(12 + 2 * k), 0, lane), { SP_R(8, 16, Y[k], rs1); split_plt<SPL>(Y[k], 1, bb); })
            B3_ADV(13 + 2 * k)
            B3_MV24(5, mma_pl(X, bb, B3_AT(13 + 2 * k), 0, lane), if (k < 3) { SP_R(0, 8, Y[k + 1 < 4 ? k + 1 : 3], rs1); split_plt<SPL>(Y[k + 1 < 4 ? k + 1 : 3], 0, ba); })
        }
        act_rows(ROW_X1, Y, SPU);
        load_bias<4>(Y, small + SM_BF, half);
        SP_R(0, 8, X[0], rs2);                                             // (the second half: behind the first chunk of the layer)
        split_plt<SPL>(X[0], 0, ba);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                               // feature_linear: chunks 20..27
            B3_ADV(20 + 2 * k)
            B3_MV24(5, mma_pl(Y, ba, B3_AT(20 + 2 * k), 0, lane), { SP_R(8, 16, X[k], rs2); split_plt<SPL>(X[k], 1, bb); })
            B3_ADV(21 + 2 * k)
            B3_MV24(5, mma_pl(Y, bb, B3_AT(21 + 2 * k), 0, lane), if (k < 3) { SP_R(0, 8, X[k + 1 < 4 ? k + 1 : 3], rs2); split_plt<SPL>(X[k + 1 < 4 ? k + 1 : 3], 0, ba); })
        }
        act_rows(ROW_X2, X, SPU);
        if constexpr (LOG2D) {   // feature_linear has no activation: its accumulators leave the plane scale here, on their way into views_linear's operand
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[t][r] *= rsF;
        }
        act_rows(ROW_Y, Y, 1.f);
#if HL_FUSE_LQ
        if constexpr (FUSE) lq_request();
#else
        if constexpr (FUSE) {   // the coarse records at the cursor and behind it: requested here, used at the end of the sample
            const float2 *p_ = vcp();
            crec = p_[64LL * (ci < Nc ? ci : Nc - 1)];
            crec1 = p_[64LL * (ci + 1 < Nc ? ci + 1 : Nc - 1)];
        }
#endif
        const float sigma_raw = dot_lane<4>(X, small + SM_AW, half) + small[SM_AB];   // X holds softplus(pts_linears.2) by now
        f32x16 V[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v4 = vinit[(t * 4 + q) * 64 + lane];
                V[t][4 * q] = v4[0]; V[t][4 * q + 1] = v4[1]; V[t][4 * q + 2] = v4[2]; V[t][4 * q + 3] = v4[3];
            }
        split_plt<SPL>(Y[0], 0, ba);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                               // views_linear (feature part): chunks 28..31, two k-groups each
            B3_ADV(28 + k)
            B3_VM(split_plt<SPL>(Y[k], 1, bb), mma_pl(V, ba, B3_AT(28 + k), 0, lane))
            B3_VM(if (k < 3) split_plt<SPL>(Y[k + 1 < 4 ? k + 1 : 3], 0, ba), mma_pl(V, bb, B3_AT(28 + k), 2, lane))
        }
        if constexpr (LOG2D) { softplus_l2_r<0, 16>(V[0], rsV); softplus_l2_r<0, 16>(V[1], rsV); }
        else { V[0] = softplus16_b3(V[0]); V[1] = softplus16_b3(V[1]); }
        act_rows(ROW_V, V, SPU);
        const float cr = dot_lane<2>(V, small + SM_RW, half) + small[SM_RB + 0];
        const float cg = dot_lane<2>(V, small + SM_RW + 64, half) + small[SM_RB + 1];
        const float cb = dot_lane<2>(V, small + SM_RW + 128, half) + small[SM_RB + 2];
#if HL_FUSE_LQ
        if constexpr (FUSE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the gathered records (and the ring's staged pair, which B3_ADV writes first thing anyway)
#endif
        B3_ADV(B3_NCH - 1)                                                          // pair 0 of the next sample (the stream of a sample is chunks 0..31)
        if constexpr (FUSE) {   // the coarse samples in front of (or at: k_composite takes the coarse one first) this depth, then the sample itself
            emit_coarse_until(zc);
            emit(zc, sigma_raw, half ? cg : cr, cb);
        } else if (tile * 32 < a.R) {   // lanes 0-31 store (sigma, r), lanes 32-63 (g, b); hidden store: see k_march
            const float2 rec = half ? make_float2(cg, cb) : make_float2(sigma_raw, cr);
            float *dst = reinterpret_cast<float *>(a.vals_out + (zt_base + 32LL * s)) + 2 * half;
            asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(dst), "v"(rec) : "memory");
        }
        zc = zn;
    }
    };
    if ((tid >> 6) < 4) body(std::false_type{});
    else body(std::true_type{});
    if constexpr (FUSE) {   // the coarse samples behind the last new depth, the last sample (distance 1e10: renderer.py:213), the image
#if HL_FUSE_LQ
        lq_request();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        {
            const float2 *p_ = vcp();
            crec = p_[64LL * (ci < Nc ? ci : Nc - 1)];
            crec1 = p_[64LL * (ci + 1 < Nc ? ci + 1 : Nc - 1)];
        }
#endif
        emit_coarse_until(3.0e38f);                                 // (zco is +inf behind the last coarse sample)
        finish_pending(1e10f);
        const float cW = __shfl(cB, lane & 31);                 // the sum of the weights (half 0 keeps it)
        if (a.flags & HL_RENDER_WHITE_BKGD) {
            const float bg = 1.f - cW;
            cA += bg;
            if (half) cB += bg;
        }
        if (a.flags & HL_RENDER_NORMALIZE_DEPTH) {
            cC = (cC - nr) / (fr_ - nr + 1e-5f);
            if (a.flags & HL_RENDER_CLAMP_DEPTH) {
                cC = cC > 1.f ? 1.f : cC;
                cC = cC < 0.f ? 0.f : cC;
            }
        }
        const int t_ = opaque_tid();
        const long long ray_ = (wg * 8 + (t_ >> 6)) * 32 + (t_ & 31);
        if (ray_ < a.R) {
            if (t_ & 32) { a.rgb[ray_ * 3 + 1] = cA; a.rgb[ray_ * 3 + 2] = cB; }
            else { a.rgb[ray_ * 3 + 0] = cA; a.acc[ray_] = cB; a.depth[ray_] = cC; }
        }
    }
#undef B3_VM
#undef B3_MV24
#undef B3_ADV
#undef B3_AT
#undef SP_R
}

// ---------------------------------------------------------------------------------------------
// importance sampling + merge: one wave per ray   [renderer.py:158-170, 533-563, 252-253]
// ---------------------------------------------------------------------------------------------
constexpr int IMP_MAX_N = 512;

struct ImpArgs {
    const float *sigma, *rays_d, *near, *far, *z, *u;
    long long R;
    int N, Ni;
    float *z_all;
    int sig_stride;   // floats between consecutive rays' sigma (1: sigma array, 4: .x of the float4 sample records)
    int new_only;     // 1: write only the n_importance new depths, sorted, [R/32][Ni][32] (k_composite merges them with the coarse ones)
    int rays_per_wave;  // 8: one workgroup per 32-ray tile; 1, 2, 4: the tile is spread over 8, 4, 2 workgroups (small ray batches)
};

__global__ __launch_bounds__(256) void k_importance(const ImpArgs a) {
    // one workgroup = one 32-ray tile (the unit the march kernels read/write as 128-byte lines); each of the 4 waves
    // runs 8 rays one after the other with wave-wide scans / searches / sort in its own LDS arrays
    __shared__ float s_wA[4][IMP_MAX_N];       // weights, then cdf
    __shared__ float s_zA[4][2 * IMP_MAX_N];   // merged depths (padded to a power of two)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *s_w = s_wA[wv], *s_z = s_zA[wv];
    const int parts = 8 / a.rays_per_wave;
    const long long tile = blockIdx.x / parts;
    const int part = (int)(blockIdx.x % parts);
    const int N = a.N, Ni = a.Ni;
    const int tot_n = N + Ni;
    int P = 1;
    while (P < tot_n) P <<= 1;
    // The densities of this wave's (up to) 8 rays at one sample share a 128-byte line of the tile-major records.  Walking the rays one after
    // the other fetched every line 8 times from HBM (24 waves per CU x 16 KB of lines do not survive in L1 / L2: 4.5 GB read per 512x512
    // view for 0.54 GB of lines): for N <= 128 a lane now fetches "its" samples of all the wave's rays up front - eight dword loads from one
    // line - and the per-ray loop takes them from registers (sg[0], shifted down after every ray).
    constexpr int PRE_B = 2;
    const bool pre = N <= 64 * PRE_B;
    float sg[8][PRE_B];
    if (pre) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int b = 0; b < PRE_B; ++b) {
                const int i = 64 * b + lane, j = wv * 8 + part * a.rays_per_wave + rr;
                sg[rr][b] = (rr < a.rays_per_wave && i < N) ? a.sigma[((tile * 32 * (long long)N + j) + 32LL * i) * a.sig_stride] : 0.f;
            }
    }
    for (int rr = 0; rr < a.rays_per_wave; ++rr) {
        const int j = wv * 8 + part * a.rays_per_wave + rr;
        const long long ray_raw = tile * 32 + j;
        const long long ray = ray_raw < a.R ? ray_raw : a.R - 1;   // padded rays recompute the last one (never read back)
        const float nr = a.near[ray], fr = a.far[ray];
        const float dxx = a.rays_d[ray * 3], dyy = a.rays_d[ray * 3 + 1], dzz = a.rays_d[ray * 3 + 2];
        const float dn = sqrtf(dxx * dxx + dyy * dyy + dzz * dzz);
        const float *sig = a.sigma + (tile * 32 * (long long)N + j) * a.sig_stride;   // [s][32] (x sig_stride)
        float *zout = a.z_all + tile * 32 * (long long)tot_n + j;           // [s][32]

        auto zval = [&](int i) -> float {
            if (a.z) return a.z[ray * N + i];
            const float t = linspace01(i, N);
            return nr * (1.f - t) + fr * t;
        };
        // weights w_i = alpha_i * prod_{k<i}(1 - alpha_k + 1e-10)
        float carry = 1.f;
        for (int base = 0; base < N; base += 64) {
            const int i = base + lane;
            float alpha = 0.f, zi = 0.f;
            if (i < N) {
                zi = zval(i);
                float dist = (i + 1 < N) ? zval(i + 1) - zi : 1e10f;
                dist = dist * dn;
                const float sraw = pre ? (base == 0 ? sg[0][0] : sg[0][1]) : sig[32LL * i * a.sig_stride];
                alpha = 1.f - expf(-softplus_exact(sraw) * dist);
                s_z[i] = zi;
            }
            const float fct = (i < N) ? (1.f - alpha + 1e-10f) : 1.f;
            const float incl = wave_incl_scan_mul(fct, lane);
            float excl = __shfl_up(incl, 1);
            if (lane == 0) excl = 1.f;
            if (i < N) s_w[i] = alpha * (carry * excl);
            carry *= __shfl(incl, 63);
        }
        __syncthreads();
        // pdf over w[1..N-2] (+1e-5), cdf[0]=0, cdf[m]=sum_{i<=m} pdf_i   (N-1 entries)
        const int M = N - 2;
        float tot = 0.f;
        for (int i = 1 + lane; i <= M; i += 64) tot += s_w[i] + 1e-5f;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d);
        __syncthreads();
        float run = 0.f;
        for (int base = 1; base <= M; base += 64) {
            const int i = base + lane;
            const float pr = (i <= M) ? (s_w[i] + 1e-5f) / tot : 0.f;
            const float incl = wave_incl_scan_add(pr, lane) + run;
            __syncthreads();
            if (i <= M) s_w[i] = incl;
            run = __shfl(incl, 63);
        }
        if (lane == 0) s_w[0] = 0.f;
        __syncthreads();
        // inverse CDF: idx = #(cdf <= u) (searchsorted right=True) over cdf[0..M]
        const int nc = M + 1;
        for (int q = lane; q < Ni; q += 64) {
            const float uq = a.u[ray * Ni + q];
            int lo = 0, hi = nc;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_w[mid] <= uq) lo = mid + 1; else hi = mid;
            }
            const int below = max(lo - 1, 0), above = min(lo, nc - 1);
            const float c0 = s_w[below], c1 = s_w[above];
            const float b0 = 0.5f * (s_z[below + 1] + s_z[below]);
            const float b1 = 0.5f * (s_z[above + 1] + s_z[above]);
            float den = c1 - c0;
            den = den < 1e-5f ? 1.f : den;
            const float t = (uq - c0) / den;
            s_z[N + q] = b0 + t * (b1 - b0);   // slots >= N: never read by the midpoint lookups above
        }
        // sort either the union (coarse + new, what the reference's torch.sort of the concatenation yields) or, for the
        // evaluate-once pipeline, only the new depths (the coarse ones are sorted already; k_composite merges)
        float *srt = a.new_only ? s_z + N : s_z;
        const int cnt = a.new_only ? Ni : tot_n;
        int Ps = 1;
        while (Ps < cnt) Ps <<= 1;
        for (int q = (a.new_only ? Ni : tot_n) + lane; q < (a.new_only ? Ps : P); q += 64) srt[q] = __builtin_inff();
        __syncthreads();
        for (int k = 2; k <= Ps; k <<= 1) {
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
                for (int e = lane; e < Ps / 2; e += 64) {
                    const int pos = 2 * jj * (e / jj) + (e % jj), par = pos + jj;
                    const bool up = (pos & k) == 0;
                    const float x = srt[pos], y = srt[par];
                    if ((x > y) == up) { srt[pos] = y; srt[par] = x; }
                }
                __syncthreads();
            }
        }
        if (a.new_only) {
            float *zn_out = a.z_all + tile * 32 * (long long)Ni + j;
            for (int i = lane; i < Ni; i += 64) zn_out[32LL * i] = srt[i];
        } else {
            for (int i = lane; i < tot_n; i += 64) zout[32LL * i] = s_z[i];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int b = 0; b < PRE_B; ++b) sg[k][b] = sg[k + 1][b];
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// merge + alpha compositing of the evaluate-once pipeline: one thread per ray walks its two sorted depth lists (coarse,
// importance) and composites the stored raw MLP outputs in merged order - the arithmetic of k_march<true>'s compositing,
// on the values k_march<true, true> stored.  [renderer.py:185-186, 213, 221-229, 252-253]
// ---------------------------------------------------------------------------------------------
struct CompArgs {
    const float *near, *far;
    const float *zc;        // coarse depths: caller rows (R, N) or null -> linspace
    const float *zn;        // new depths, sorted, tile-major [R/32][Ni][32]
    const float4 *vc, *vn;  // raw (sigma, r, g, b), tile-major [R/32][N][32] and [R/32][Ni][32]
    long long R;
    int N, Ni;
    unsigned flags;
    float *rgb, *acc, *depth;
    const float *noise;     // training: added to sigma_raw of sorted sample s of ray r, rows (R, N+Ni)  [renderer.py:212]; or null
};

__global__ __launch_bounds__(256) void k_composite(const CompArgs a) {
    const long long ray = (long long)blockIdx.x * 256 + threadIdx.x;
    if (ray >= a.R) return;
    const long long tile = ray >> 5;
    const int r = (int)(ray & 31);
    const int N = a.N, Ni = a.Ni, S = N + Ni;
    const float nr = a.near[ray], fr = a.far[ray];
    const float4 *vc = a.vc + tile * 32 * (long long)N + r, *vn = a.vn + tile * 32 * (long long)Ni + r;
    const float *zn = a.zn + tile * 32 * (long long)Ni + r;
    auto zcoarse = [&](int i) -> float {
        if (a.zc) return a.zc[ray * N + i];
        const float t = linspace01(i, N);
        return nr * (1.f - t) + fr * t;
    };
    const float inf = __builtin_inff();
    int ia = 0, ib = 0;
    float za = zcoarse(0), zb = zn[0];
    // current element
    bool fromA = za <= zb;
    float zcur = fromA ? za : zb;
    float4 vcur = fromA ? vc[0] : vn[0];
    if (fromA) { ++ia; za = ia < N ? zcoarse(ia) : inf; } else { ++ib; zb = ib < Ni ? zn[32LL * ib] : inf; }
    float T = 1.f, acc_w = 0.f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f;
    for (int s = 0; s < S; ++s) {
        float znext = 0.f;
        float4 vnext = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s + 1 < S) {
            const bool nA = za <= zb;
            znext = nA ? za : zb;
            vnext = nA ? vc[32LL * ia] : vn[32LL * ib];
            if (nA) { ++ia; za = ia < N ? zcoarse(ia) : inf; } else { ++ib; zb = ib < Ni ? zn[32LL * ib] : inf; }
        }
        const float dist = (s + 1 < S) ? znext - zcur : 1e10f;
        const float sraw = a.noise ? vcur.x + a.noise[ray * S + s] : vcur.x;
        // (training noise: the exact forms of rounds 1-5, what k_composite_bwd differentiates; inference: the shared comp_* forms)
        const float alpha = a.noise ? 1.f - expf(-softplus_exact(sraw) * dist) : (s + 1 < S ? comp_alpha(sraw, dist) : comp_alpha_last(sraw));
        const float w = alpha * T;
        acc_w += w;
        acc_r += (a.noise ? 1.f / (1.f + expf(-vcur.y)) : comp_sigmoid(vcur.y)) * w;
        acc_g += (a.noise ? 1.f / (1.f + expf(-vcur.z)) : comp_sigmoid(vcur.z)) * w;
        acc_b += (a.noise ? 1.f / (1.f + expf(-vcur.w)) : comp_sigmoid(vcur.w)) * w;
        acc_d += w * zcur;
        T *= (1.f - alpha + 1e-7f);
        zcur = znext;
        vcur = vnext;
    }
    if (a.flags & HL_RENDER_WHITE_BKGD) {
        const float bg = 1.f - acc_w;
        acc_r += bg; acc_g += bg; acc_b += bg;
    }
    if (a.flags & HL_RENDER_NORMALIZE_DEPTH) {
        acc_d = (acc_d - nr) / (fr - nr + 1e-5f);
        if (a.flags & HL_RENDER_CLAMP_DEPTH) {
            acc_d = acc_d > 1.f ? 1.f : acc_d;
            acc_d = acc_d < 0.f ? 0.f : acc_d;
        }
    }
    a.rgb[ray * 3 + 0] = acc_r;
    a.rgb[ray * 3 + 1] = acc_g;
    a.rgb[ray * 3 + 2] = acc_b;
    a.acc[ray] = acc_w;
    a.depth[ray] = acc_d;
}

// ---------------------------------------------------------------------------------------------
// canonical-space deformation   [renderer.py:52-132 deform_target2c / deform_target2c_op]
// Everything the reference does after the 1-NN lookup depends on the query only through the id of the nearest body vertex
// (blend weights, both blended joint transforms, the three blend-shape offsets), so the host mirror folds it into ONE table
// row per vertex and this kernel is: world -> SMPL space, brute-force nearest vertex (vertices staged through LDS as float4,
// squared distance as ((dx*dx + dy*dy) + dz*dz), first index wins ties), then the row's operations in the reference's order:
//     can = Rinv (q - t);  can -= pose_off;  can -= shape_off;  can += pose_off_big;  can = Rbig can + tbig
//     dir = Rbig (Rinv dir_smpl)
// Table row (36 floats): t[3] Rinv[9] pose_off[3] shape_off[3] pose_off_big[3] Rbig[9] tbig[3] pad[3].
// ---------------------------------------------------------------------------------------------
struct DeformCommon {
    float R[9], Th[3];            // world -> SMPL space: (p - Th) R
    const float4 *verts;          // (V) SMPL-space body vertices, w unused
    const float *table;           // (V,36)
    int V;
};
struct DeformArgs {
    DeformCommon c;
    const float *pts, *dirs;      // (P,3) world space; dirs may be null
    long long P;
    float *can_pts, *can_dirs;    // (P,3)
    int *vid;                     // (P) nearest vertex ids or null
};
struct DeformRaysArgs {
    DeformCommon c;
    const float *rays_o, *rays_d, *near, *far;
    const float *z;               // null (linspace), caller rows (R,S) or tile-major [R/32][S][32]
    int z_tiled;
    long long R;
    int S;
    float4 *pts_c, *dirs_c;       // tile-major [R/32][S][32]
    unsigned long long *counter;  // k_deform_rays_cull: work counter
};

__device__ __forceinline__ void mat3_apply(const float *m, float x, float y, float z, float &ox, float &oy, float &oz) {
    ox = (m[0] * x + m[1] * y) + m[2] * z;
    oy = (m[3] * x + m[4] * y) + m[5] * z;
    oz = (m[6] * x + m[7] * y) + m[8] * z;
}

constexpr int DEFORM_TILE = 2048;   // vertices per LDS tile (32 KB)

// all 256 threads of the workgroup must call this (the vertex tiles are staged cooperatively); (wx,wy,wz) world-space query,
// (ex,ey,ez) world-space direction (ignored unless with_dir).  Returns the nearest vertex id.
__device__ __forceinline__ int deform_query(const DeformCommon &c, float4 *sv, float wx, float wy, float wz, bool with_dir, float ex,
                                            float ey, float ez, float (&cp)[3], float (&cd)[3]) {
    wx -= c.Th[0]; wy -= c.Th[1]; wz -= c.Th[2];
    // (p - Th) R: row vector times matrix
    const float qx = (wx * c.R[0] + wy * c.R[3]) + wz * c.R[6];
    const float qy = (wx * c.R[1] + wy * c.R[4]) + wz * c.R[7];
    const float qz = (wx * c.R[2] + wy * c.R[5]) + wz * c.R[8];
    float best = 3.0e38f;
    int bid = 0;
    for (int v0 = 0; v0 < c.V; v0 += DEFORM_TILE) {
        const int n = min(DEFORM_TILE, c.V - v0);
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += 256) sv[k] = c.verts[v0 + k];
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < n; ++k) {
            const float4 v = sv[k];                       // same address for the whole wave: LDS broadcast
            const float dx = qx - v.x, dy = qy - v.y, dz = qz - v.z;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d < best) { best = d; bid = v0 + k; }
        }
    }
    // the row is fetched as 9 x 16 bytes (rows are 144 B apart, 16-byte aligned): 36 separate dword gathers, each touching up to 64
    // different lines per wave, cost more than the culled nearest-vertex search
    float row[36];
    {
        const float4 *row4 = reinterpret_cast<const float4 *>(c.table + (long long)bid * 36);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float4 t = row4[i];
            row[4 * i + 0] = t.x; row[4 * i + 1] = t.y; row[4 * i + 2] = t.z; row[4 * i + 3] = t.w;
        }
    }
    float cx, cy, cz;
    mat3_apply(row + 3, qx - row[0], qy - row[1], qz - row[2], cx, cy, cz);
    cx -= row[12]; cy -= row[13]; cz -= row[14];
    cx -= row[15]; cy -= row[16]; cz -= row[17];
    cx += row[18]; cy += row[19]; cz += row[20];
    float ox, oy, oz;
    mat3_apply(row + 21, cx, cy, cz, ox, oy, oz);
    cp[0] = ox + row[30]; cp[1] = oy + row[31]; cp[2] = oz + row[32];
    if (with_dir) {
        ex -= c.Th[0]; ey -= c.Th[1]; ez -= c.Th[2];      // the reference subtracts Th from directions too (:128)
        const float sx = (ex * c.R[0] + ey * c.R[3]) + ez * c.R[6];
        const float sy = (ex * c.R[1] + ey * c.R[4]) + ez * c.R[7];
        const float sz = (ex * c.R[2] + ey * c.R[5]) + ez * c.R[8];
        float tx, ty, tz;
        mat3_apply(row + 3, sx, sy, sz, tx, ty, tz);
        mat3_apply(row + 21, tx, ty, tz, cd[0], cd[1], cd[2]);
    }
    return bid;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Nearest body vertex for TWO queries per thread (packed fp32 math: one v_pk_* instruction serves both queries, and one LDS
// broadcast read of a vertex serves 128 queries per wave).  Squared distance as fma(dz,dz, fma(dy,dy, dx*dx)).  Two levels: the
// scan only keeps the minimum DISTANCE per chunk of 32 vertices (one v_min per query and vertex instead of a compare and two
// selects) and the first chunk that holds the overall minimum; the winning chunk is then re-scanned for the first vertex at
// exactly that distance - the same vertex a flat first-index-wins scan returns.
constexpr int DEFORM_CHUNK = 32;
__device__ __forceinline__ void nearest2(const DeformCommon &c, float4 *sv, f32x2 qx, f32x2 qy, f32x2 qz, int (&bid)[2]) {
    f32x2 best = {3.0e38f, 3.0e38f};
    int bch[2] = {0, 0};
    for (int v0 = 0; v0 < c.V; v0 += DEFORM_TILE) {
        const int n = min(DEFORM_TILE, c.V - v0);
        __syncthreads();
        for (int k = threadIdx.x; k < DEFORM_TILE; k += 256)       // pad the last chunk with far-away points
            sv[k] = k < n ? c.verts[v0 + k] : make_float4(1.0e18f, 1.0e18f, 1.0e18f, 0.f);
        __syncthreads();
        for (int k0 = 0; k0 < n; k0 += DEFORM_CHUNK) {
            f32x2 m = {3.0e38f, 3.0e38f};
#pragma unroll
            for (int k = 0; k < DEFORM_CHUNK; ++k) {
                const float4 v = sv[k0 + k];
                const f32x2 dx = qx - v.x, dy = qy - v.y, dz = qz - v.z;
                const f32x2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                m = __builtin_elementwise_min(m, d);
            }
            if (m[0] < best[0]) { best[0] = m[0]; bch[0] = v0 + k0; }
            if (m[1] < best[1]) { best[1] = m[1]; bch[1] = v0 + k0; }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        bid[q] = bch[q];
        const int nn = min(DEFORM_CHUNK, c.V - bch[q]);
        for (int k = nn - 1; k >= 0; --k) {                         // descending: the lowest matching index is written last
            const float4 v = c.verts[bch[q] + k];
            const float dx = qx[q] - v.x, dy = qy[q] - v.y, dz = qz[q] - v.z;
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (d == best[q]) bid[q] = bch[q] + k;
        }
    }
}

// the table row of vertex `bid` applied to an SMPL-space query / direction, in the reference's order
__device__ __forceinline__ void apply_row(const DeformCommon &c, int bid, float qx, float qy, float qz, bool with_dir, float sx, float sy,
                                          float sz, float (&cp)[3], float (&cd)[3]) {
    // the row is fetched as 9 x 16 bytes (rows are 144 B apart, 16-byte aligned): 36 separate dword gathers, each touching up to 64
    // different lines per wave, cost more than the culled nearest-vertex search
    float row[36];
    {
        const float4 *row4 = reinterpret_cast<const float4 *>(c.table + (long long)bid * 36);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float4 t = row4[i];
            row[4 * i + 0] = t.x; row[4 * i + 1] = t.y; row[4 * i + 2] = t.z; row[4 * i + 3] = t.w;
        }
    }
    float cx, cy, cz;
    mat3_apply(row + 3, qx - row[0], qy - row[1], qz - row[2], cx, cy, cz);
    cx -= row[12]; cy -= row[13]; cz -= row[14];
    cx -= row[15]; cy -= row[16]; cz -= row[17];
    cx += row[18]; cy += row[19]; cz += row[20];
    float ox, oy, oz;
    mat3_apply(row + 21, cx, cy, cz, ox, oy, oz);
    cp[0] = ox + row[30]; cp[1] = oy + row[31]; cp[2] = oz + row[32];
    if (with_dir) {
        float tx, ty, tz;
        mat3_apply(row + 3, sx, sy, sz, tx, ty, tz);
        mat3_apply(row + 21, tx, ty, tz, cd[0], cd[1], cd[2]);
    }
}

__global__ __launch_bounds__(256) void k_deform_points(const DeformArgs a) {
    __shared__ float4 sv[DEFORM_TILE];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < a.P;
    const long long ic = live ? i : a.P - 1;
    float cp[3], cd[3];
    const bool wd = a.dirs != nullptr;
    const int bid = deform_query(a.c, sv, a.pts[ic * 3 + 0], a.pts[ic * 3 + 1], a.pts[ic * 3 + 2], wd, wd ? a.dirs[ic * 3 + 0] : 0.f,
                                 wd ? a.dirs[ic * 3 + 1] : 0.f, wd ? a.dirs[ic * 3 + 2] : 0.f, cp, cd);
    if (!live) return;
    a.can_pts[i * 3 + 0] = cp[0]; a.can_pts[i * 3 + 1] = cp[1]; a.can_pts[i * 3 + 2] = cp[2];
    if (a.vid) a.vid[i] = bid;
    if (wd) { a.can_dirs[i * 3 + 0] = cd[0]; a.can_dirs[i * 3 + 1] = cd[1]; a.can_dirs[i * 3 + 2] = cd[2]; }
}

// sample points of a batch of rays (o + d*z, the unit ray direction as view direction: renderer.py:192, 258-259) deformed into
// canonical space, written tile-major for k_march<.., POINTS>: thread = (tile, sample, ray-in-tile), rays fastest
__global__ __launch_bounds__(256) void k_deform_rays(const DeformRaysArgs a) {
    __shared__ float4 sv[DEFORM_TILE];
    // thread = (tile, sample pair, ray-in-tile), rays fastest; the pair is samples (s, s + ceil(S/2)) of the same ray
    const long long tiles_n = (a.R + 31) / 32;
    const int SH = (a.S + 1) / 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, total = tiles_n * 32 * SH;
    const long long ic = idx < total ? idx : total - 1;
    const long long tile = ic / (32LL * SH);
    const int rem = (int)(ic - tile * 32LL * SH), s0 = rem >> 5, r = rem & 31;
    const int s1 = min(s0 + SH, a.S - 1);                           // odd S: the last thread row repeats the final sample
    const long long ray_raw = tile * 32 + r, ray = ray_raw < a.R ? ray_raw : a.R - 1;
    const float ox = a.rays_o[ray * 3 + 0], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
    const float dx = a.rays_d[ray * 3 + 0], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
    auto depth = [&](int s) -> float {
        if (a.z) return a.z_tiled ? a.z[(tile * a.S + s) * 32 + r] : a.z[ray * a.S + s];
        const float t = linspace01(s, a.S);
        return a.near[ray] * (1.f - t) + a.far[ray] * t;
    };
    const DeformCommon &c = a.c;
    float q[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float zc = depth(k ? s1 : s0);
        const float wx = (ox + dx * zc) - c.Th[0], wy = (oy + dy * zc) - c.Th[1], wz = (oz + dz * zc) - c.Th[2];
        q[k][0] = (wx * c.R[0] + wy * c.R[3]) + wz * c.R[6];       // (p - Th) R
        q[k][1] = (wx * c.R[1] + wy * c.R[4]) + wz * c.R[7];
        q[k][2] = (wx * c.R[2] + wy * c.R[5]) + wz * c.R[8];
    }
    int bid[2];
    nearest2(c, sv, f32x2{q[0][0], q[1][0]}, f32x2{q[0][1], q[1][1]}, f32x2{q[0][2], q[1][2]}, bid);
    if (idx >= total) return;
    // unit ray direction as view direction (renderer.py:258-259), through the same world -> SMPL map as the points (:128)
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float ex = dx / nrm - c.Th[0], ey = dy / nrm - c.Th[1], ez = dz / nrm - c.Th[2];
    const float sx = (ex * c.R[0] + ey * c.R[3]) + ez * c.R[6];
    const float sy = (ex * c.R[1] + ey * c.R[4]) + ez * c.R[7];
    const float sz = (ex * c.R[2] + ey * c.R[5]) + ez * c.R[8];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int s = k ? s1 : s0;
        float cp[3], cd[3];
        apply_row(c, bid[k], q[k][0], q[k][1], q[k][2], true, sx, sy, sz, cp, cd);
        const long long o = (tile * a.S + s) * 32 + r;
        a.pts_c[o] = make_float4(cp[0], cp[1], cp[2], 0.f);
        a.dirs_c[o] = make_float4(cd[0], cd[1], cd[2], 0.f);
    }
}

// k_deform_rays with group culling (exact): a wave takes the 64 sample points (16 neighbouring rays x 4 consecutive depths) of one
// step, which lie within a few centimetres of each other, and brute-forces only the vertices that can be nearest to ANY of them:
// with m the centroid of the 64 points, rho their largest distance to m and U an upper bound of m's distance to the body (distance
// to some actual vertices), the nearest vertex v* of a point p of the group obeys |m - v*| <= |p - v*| + rho <= |p - v_m| + rho <=
// U + 2 rho.  Vertices are screened in index order in two levels - bounding spheres of runs of 64 consecutive vertices (meshes are
// index-local; costs nothing when they are not), then the vertices of the surviving runs - and appended to a per-wave list in LDS
// that the 64 lanes then scan with the same distance expression and strict "<" as the full scan, so the result (first index wins
// ties) is the full scan's.  The list is flushed whenever it fills up, so no group is too large.
constexpr int DC_MAXV = 7168, DC_WAVES = 8, DC_LIST = 256, DC_SLOTS = DC_LIST / 2 + 4;   // list entries; slots incl. the padding of a flush
__global__ __launch_bounds__(DC_WAVES * 64) void k_deform_rays_cull(const DeformRaysArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dc_lds[];
    float4 *sv = reinterpret_cast<float4 *>(dc_lds);                 // [V] vertices
    float4 *scl = sv + DC_MAXV;                                      // [V/64] bounding spheres of vertex runs
    // this wave's candidate list, two entries per slot so that the scan runs on packed fp32: LA = (x0, x1, y0, y1), LB = (z0, z1, id0, id1)
    float4 *LA = scl + DC_MAXV / 64 + (threadIdx.x >> 6) * DC_SLOTS * 2, *LB = LA + DC_SLOTS;
    const DeformCommon &c = a.c;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int V = c.V, NC = (V + 63) / 64;
    for (int i = threadIdx.x; i < NC * 64; i += DC_WAVES * 64) sv[i] = i < V ? c.verts[i] : make_float4(1.0e18f, 1.0e18f, 1.0e18f, 0.f);
    __syncthreads();
    auto wsum = [](float v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d); return v; };
    auto wmax = [](float v) { for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d)); return v; };
    auto wmin = [](float v) { for (int d = 32; d > 0; d >>= 1) v = fminf(v, __shfl_xor(v, d)); return v; };
    for (int k = wv; k < NC; k += DC_WAVES) {                        // bounding sphere of vertices 64k .. 64k+63
        const bool on = 64 * k + lane < V;
        const float4 v = sv[64 * k + lane];
        const float n = wsum(on ? 1.f : 0.f);
        const float cx = wsum(on ? v.x : 0.f) / n, cy = wsum(on ? v.y : 0.f) / n, cz = wsum(on ? v.z : 0.f) / n;
        const float dx = v.x - cx, dy = v.y - cy, dz = v.z - cz;
        const float r = wmax(on ? sqrtf(dx * dx + dy * dy + dz * dz) : 0.f);
        if (lane == 0) scl[k] = make_float4(cx, cy, cz, r * 1.0001f + 1e-6f);
    }
    __syncthreads();
    const long long tiles_n = (a.R + 31) / 32;
    const int SP = (a.S + 3) / 4;                                    // a step: 16 neighbouring rays x depths 4 j .. 4 j + 3
    const long long items = tiles_n * 2 * SP;
    // unit ray direction as view direction (renderer.py:258-259), through the same world -> SMPL map as the points (:128)
    // Steps along the silhouette (groups that mix rays through the body with rays that miss the box) keep every vertex and cost ~50x
    // the others; they come in runs (a tile's 2 x SP steps, the tiles of an image column), so the steps are dealt to the waves through a
    // multiplicative hash (a bijection on [0, M), M the next power of two) instead of round-robin.
    unsigned long long M = 1;
    while (M < (unsigned long long)items) M <<= 1;
    for (;;) {
        // dynamic: the next step off a global counter (zeroed by the launcher) - the expensive steps are too uneven for a static deal
        unsigned long long it = 0;
        if (lane == 0) it = atomicAdd(a.counter, 1ull);
        it = __shfl(it, 0);
        if (it >= M) break;
        const long long item = (long long)((it * 0x9E3779B97F4A7C15ull) & (M - 1));
        if (item >= items) continue;
        const long long tile = item / (2 * SP);
        const int sub = (int)(item - tile * 2 * SP);
        const int s_raw = 4 * (sub >> 1) + (lane >> 4), r = 16 * (sub & 1) + (lane & 15);
        const int s = s_raw < a.S ? s_raw : a.S - 1;
        const long long ray_raw = tile * 32 + r, ray = ray_raw < a.R ? ray_raw : a.R - 1;
        const float ox = a.rays_o[ray * 3 + 0], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
        const float dx = a.rays_d[ray * 3 + 0], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
        float zc;
        if (a.z) zc = a.z_tiled ? a.z[(tile * a.S + s) * 32 + r] : a.z[ray * a.S + s];
        else { const float t = linspace01(s, a.S); zc = a.near[ray] * (1.f - t) + a.far[ray] * t; }
        const float wx = (ox + dx * zc) - c.Th[0], wy = (oy + dy * zc) - c.Th[1], wz = (oz + dz * zc) - c.Th[2];
        const float qx = (wx * c.R[0] + wy * c.R[3]) + wz * c.R[6];       // (p - Th) R
        const float qy = (wx * c.R[1] + wy * c.R[4]) + wz * c.R[7];
        const float qz = (wx * c.R[2] + wy * c.R[5]) + wz * c.R[8];
        // group geometry
        const float mx = wsum(qx) * (1.f / 64.f), my = wsum(qy) * (1.f / 64.f), mz = wsum(qz) * (1.f / 64.f);
        const float rho = wmax(sqrtf((qx - mx) * (qx - mx) + (qy - my) * (qy - my) + (qz - mz) * (qz - mz))) * 1.0001f + 1e-6f;
        // U: m's distance to the body - one lane-parallel pass over the vertices (64 per step, loads independent of each other)
        float u2 = 3.0e38f;
#pragma unroll 4
        for (int k = 0; k < NC; ++k) {
            const float4 v = sv[64 * k + lane];
            u2 = fminf(u2, (v.x - mx) * (v.x - mx) + (v.y - my) * (v.y - my) + (v.z - mz) * (v.z - mz));
        }
        const float lim = sqrtf(wmin(u2)) * 1.0001f + 2.f * rho + 1e-6f;     // |m - v*| <= lim for every point of the group
        const float lim2 = lim * lim * 1.0001f;
        // runs whose bounding sphere reaches into the ball
        unsigned long long runs[(DC_MAXV / 64 + 63) / 64];
#pragma unroll
        for (int w = 0; w < (DC_MAXV / 64 + 63) / 64; ++w) {
            const int k = 64 * w + lane;
            bool hit = false;
            if (k < NC) {
                const float4 b = scl[k];
                const float d = sqrtf((b.x - mx) * (b.x - mx) + (b.y - my) * (b.y - my) + (b.z - mz) * (b.z - mz));
                hit = d - b.w <= lim;
            }
            runs[w] = __ballot(hit);
        }
        float best = 3.0e38f;
        int bid = 0;
        int fill = 0;
        const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        auto pair_d = [&](int slot) -> f32x2 {       // squared distances to the two entries of a slot: the full scan's expression (nearest2)
            const float4 A = LA[slot], B = LB[slot];
            const f32x2 ddx = qx2 - f32x2{A.x, A.y}, ddy = qy2 - f32x2{A.z, A.w}, ddz = qz2 - f32x2{B.x, B.y};
            return __builtin_elementwise_fma(ddz, ddz, __builtin_elementwise_fma(ddy, ddy, ddx * ddx));
        };
        auto put = [&](int pos, float x, float y, float z, int id) {
            float *pa = reinterpret_cast<float *>(LA + (pos >> 1)) + (pos & 1), *pb = reinterpret_cast<float *>(LB + (pos >> 1)) + (pos & 1);
            pa[0] = x; pa[2] = y; pb[0] = z; pb[2] = __builtin_bit_cast(float, id);
        };
        auto flush = [&]() {
            if (lane < 8) put(fill + lane, 1.0e18f, 1.0e18f, 1.0e18f, 0);       // pad to whole groups of 4 slots
            const int nslots = (fill + 1) >> 1;
            int won = -1;                                                       // first slot of the group that improved `best`
            for (int i0 = 0; i0 < nslots; i0 += 4) {
                f32x2 m = __builtin_elementwise_min(__builtin_elementwise_min(pair_d(i0), pair_d(i0 + 1)),
                                                    __builtin_elementwise_min(pair_d(i0 + 2), pair_d(i0 + 3)));
                const float cm = fminf(m[0], m[1]);
                if (cm < best) { best = cm; won = i0; }
            }
            if (won >= 0) {
                for (int j = 3; j >= 0; --j) {                                   // descending: the lowest index is written last
                    const f32x2 d = pair_d(won + j);
                    const float4 B = LB[won + j];
                    if (d[1] == best) bid = __builtin_bit_cast(int, B.w);
                    if (d[0] == best) bid = __builtin_bit_cast(int, B.z);
                }
            }
            fill = 0;
        };
#pragma unroll
        for (int w = 0; w < (DC_MAXV / 64 + 63) / 64; ++w) {
            unsigned long long mask = runs[w];
            while (mask) {
                const int k = 64 * w + __builtin_ctzll(mask);
                mask &= mask - 1;
                const int vi = 64 * k + lane;
                const float4 v = sv[vi];
                const float d2 = (v.x - mx) * (v.x - mx) + (v.y - my) * (v.y - my) + (v.z - mz) * (v.z - mz);
                const bool keep = vi < V && d2 <= lim2;
                const unsigned long long km = __ballot(keep);
                if (keep) put(fill + __builtin_popcountll(km & ((1ull << lane) - 1ull)), v.x, v.y, v.z, vi);
                fill += __builtin_popcountll(km);
                if (fill > DC_LIST - 64) flush();
            }
        }
        flush();
        if (ray_raw < a.R || true) {       // padding rays of the last tile are written too (finite values for the march kernel)
            const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
            const float ex = dx / nrm - c.Th[0], ey = dy / nrm - c.Th[1], ez = dz / nrm - c.Th[2];
            const float sx = (ex * c.R[0] + ey * c.R[3]) + ez * c.R[6];
            const float sy = (ex * c.R[1] + ey * c.R[4]) + ez * c.R[7];
            const float sz = (ex * c.R[2] + ey * c.R[5]) + ez * c.R[8];
            float cp[3], cd[3];
            apply_row(c, bid, qx, qy, qz, true, sx, sy, sz, cp, cd);
            if (s_raw < a.S) {
                const long long o = (tile * a.S + s) * 32 + r;
                a.pts_c[o] = make_float4(cp[0], cp[1], cp[2], 0.f);
                a.dirs_c[o] = make_float4(cd[0], cd[1], cd[2], 0.f);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-view ray generation   [SynBodyView_datasets.py:316-329 get_rays, :370-403 get_near_far, :422-433]
// One thread per pixel, float64 like the reference's numpy (K, R, T are float64 there), rounded to float32 exactly
// where sample_ray_batch casts.  Term order follows oracle/camera_oracle.py (no FMA contraction in this build).
// ---------------------------------------------------------------------------------------------
struct CamArgs {
    double Ki[9], R[9], T[3], o[3];   // inv(K), world->camera rotation, translation, camera centre -(R^T T)
    double b[6];                      // padded bounds: min xyz, max xyz
    int H, W;
    float *rays_o, *rays_d, *near, *far;
    unsigned char *mask;
};

__global__ __launch_bounds__(256) void k_camera_rays(const CamArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.H * a.W) return;
    const double x = (double)(int)(i % a.W), y = (double)(int)(i / a.W);
    double pc[3], q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pc[c] = (x * a.Ki[c * 3 + 0] + y * a.Ki[c * 3 + 1]) + a.Ki[c * 3 + 2];
        q[c] = pc[c] - a.T[c];
    }
    float of[3], df[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double pw = (q[0] * a.R[0 * 3 + c] + q[1] * a.R[1 * 3 + c]) + q[2] * a.R[2 * 3 + c];
        df[c] = (float)(pw - a.o[c]);
        of[c] = (float)a.o[c];
        if (df[c] == 0.0f) df[c] = 1e-8f;   // get_near_far writes this into the caller's ray_d
    }
    const double o[3] = {(double)of[0], (double)of[1], (double)of[2]};
    const double d[3] = {(double)df[0], (double)df[1], (double)df[2]};
    const float norm32 = sqrtf((df[0] * df[0] + df[1] * df[1]) + df[2] * df[2]);
    const double eps = 1e-6;
    int cnt = 0;
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {   // min_x, min_y, min_z, max_x, max_y, max_z
        const int ax = k % 3;
        const double t = (a.b[k] - o[ax]) / d[ax];
        const double p0 = t * d[0] + o[0], p1 = t * d[1] + o[1], p2 = t * d[2] + o[2];
        const bool inside = p0 >= a.b[0] - eps && p0 <= a.b[3] + eps && p1 >= a.b[1] - eps && p1 <= a.b[4] + eps &&
                            p2 >= a.b[2] - eps && p2 <= a.b[5] + eps;
        if (inside) {
            const double e0 = p0 - o[0], e1 = p1 - o[1], e2 = p2 - o[2];
            const double r = sqrt((e0 * e0 + e1 * e1) + e2 * e2) / (double)norm32;
            if (cnt == 0) d0 = r;
            else if (cnt == 1) d1 = r;
            ++cnt;
        }
    }
    const bool hit = cnt == 2;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.rays_o[i * 3 + c] = of[c];
        a.rays_d[i * 3 + c] = df[c];
    }
    a.near[i] = hit ? (float)fmin(d0, d1) : 0.f;
    a.far[i] = hit ? (float)fmax(d0, d1) : 1.f;
    if (a.mask) a.mask[i] = hit ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Training: backward of render_core for the tri-plane fitting loop
// (recon_NeRF/run_nerf_batch.py:236-265; human_diffusion/NeRF/renderer.py:172-231 with test=False).  The importance depths are
// drawn under no_grad in the reference (renderer.py:243-253), so the gradient flows only through the evaluation of the 2N sample
// points and the compositing:
//   k_composite_bwd   dL/d(rgb, acc) per ray -> dL/d(sigma_raw, rgb_raw) per sample
//   k_mlp_bwd         -> layer deltas (for the weight gradients) and, through the bilinear taps, the tri-plane gradient
// ---------------------------------------------------------------------------------------------
struct CompBwdArgs {
    CompArgs c;
    const float *g_rgb, *g_acc;   // (R,3), (R)
    float4 *dvc, *dvn;            // out, same layout as c.vc / c.vn
    float *sT;                    // scratch [R/32][N+Ni][32]: transmittance in front of sorted sample s
    int *sSrc;                    // scratch: which list / index sorted sample s came from
    float *del;                   // delta matrix: rows DROW_REC..+3 receive the same four values per sample point (row layout, for
    long long del_stride;         //   the alpha_linear / rgb_linear weight gradients); columns: coarse pass first, then the new depths
};

// w_s = alpha_s T_s, T_{s+1} = T_s (1 - alpha_s + 1e-7), L = sum_s gw_s w_s with gw_s = g_rgb . c_s + g_acc.
// dL/dalpha_s = T_s (gw_s - Q_s),  Q_s = sum_{k>s} gw_k alpha_k prod_{s<j<k} (1 - alpha_j + 1e-7),  Q_{s-1} = gw_s alpha_s + (1 - alpha_s + 1e-7) Q_s:
// one forward walk (records T_s and the merge order), one backward walk; no division by the transmittance.
__global__ __launch_bounds__(256) void k_composite_bwd(const CompBwdArgs b) {
    const CompArgs &a = b.c;
    const long long ray = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long tiles_n = (a.R + 31) / 32;
    if (ray >= tiles_n * 32) return;
    const long long tile = ray >> 5;
    const int r = (int)(ray & 31);
    const int N = a.N, Ni = a.Ni, S = N + Ni;
    float4 *dvc = b.dvc + tile * 32 * (long long)N + r, *dvn = b.dvn + tile * 32 * (long long)Ni + r;
    float *drow_c = b.del + (long long)DROW_REC * b.del_stride + tile * 32 * (long long)N + r;
    float *drow_n = b.del + (long long)DROW_REC * b.del_stride + tiles_n * 32 * (long long)N + tile * 32 * (long long)Ni + r;
    auto put = [&](float4 *rec, float *row, int i, const float4 d) {
        rec[32LL * i] = d;
        row[32LL * i] = d.x;
        row[32LL * i + b.del_stride] = d.y;
        row[32LL * i + 2 * b.del_stride] = d.z;
        row[32LL * i + 3 * b.del_stride] = d.w;
    };
    if (ray >= a.R) {   // padding rays of the last tile: zero deltas, so that they drop out of every reduction over sample points
        for (int i = 0; i < N; ++i) put(dvc, drow_c, i, make_float4(0.f, 0.f, 0.f, 0.f));
        for (int i = 0; i < Ni; ++i) put(dvn, drow_n, i, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    const float nr = a.near[ray], fr = a.far[ray];
    const float4 *vc = a.vc + tile * 32 * (long long)N + r, *vn = a.vn + tile * 32 * (long long)Ni + r;
    const float *zn = a.zn + tile * 32 * (long long)Ni + r;
    float *sT = b.sT + tile * 32 * (long long)S + r;
    int *sSrc = b.sSrc + tile * 32 * (long long)S + r;
    auto zcoarse = [&](int i) -> float {
        if (a.zc) return a.zc[ray * N + i];
        const float t = linspace01(i, N);
        return nr * (1.f - t) + fr * t;
    };
    auto sraw_of = [&](const float4 &v, int s) -> float { return a.noise ? v.x + a.noise[ray * S + s] : v.x; };
    const float inf = __builtin_inff();
    {   // forward walk: same merge and arithmetic as k_composite
        int ia = 0, ib = 0;
        float za = zcoarse(0), zb = zn[0];
        bool fromA = za <= zb;
        float zcur = fromA ? za : zb;
        float4 vcur = fromA ? vc[0] : vn[0];
        int code = fromA ? 0 : (int)0x80000000;
        if (fromA) { ++ia; za = ia < N ? zcoarse(ia) : inf; } else { ++ib; zb = ib < Ni ? zn[32LL * ib] : inf; }
        float T = 1.f;
        for (int s = 0; s < S; ++s) {
            float znext = 0.f;
            float4 vnext = make_float4(0.f, 0.f, 0.f, 0.f);
            int cnext = 0;
            if (s + 1 < S) {
                const bool nA = za <= zb;
                znext = nA ? za : zb;
                vnext = nA ? vc[32LL * ia] : vn[32LL * ib];
                cnext = nA ? ia : (ib | (int)0x80000000);
                if (nA) { ++ia; za = ia < N ? zcoarse(ia) : inf; } else { ++ib; zb = ib < Ni ? zn[32LL * ib] : inf; }
            }
            const float dist = (s + 1 < S) ? znext - zcur : 1e10f;
            const float alpha = 1.f - expf(-softplus_exact(sraw_of(vcur, s)) * dist);
            sT[32LL * s] = T;
            sSrc[32LL * s] = code;
            T *= (1.f - alpha + 1e-7f);
            zcur = znext; vcur = vnext; code = cnext;
        }
    }
    const float gr = b.g_rgb[ray * 3 + 0], gg = b.g_rgb[ray * 3 + 1], gb = b.g_rgb[ray * 3 + 2];
    const float ga = b.g_acc[ray] - ((a.flags & HL_RENDER_WHITE_BKGD) ? gr + gg + gb : 0.f);   // rgb += 1 - acc
    float Q = 0.f, znext = 0.f;
    for (int s = S - 1; s >= 0; --s) {
        const int code = sSrc[32LL * s];
        const bool fromB = code < 0;
        const int idx = code & 0x7fffffff;
        const float T = sT[32LL * s];
        const float zcur = fromB ? zn[32LL * idx] : zcoarse(idx);
        const float4 v = fromB ? vn[32LL * idx] : vc[32LL * idx];
        const float dist = (s + 1 < S) ? znext - zcur : 1e10f;
        const float x = sraw_of(v, s);
        const float e = expf(-softplus_exact(x) * dist);
        const float alpha = 1.f - e;
        const float cr = 1.f / (1.f + expf(-v.y)), cg = 1.f / (1.f + expf(-v.z)), cb = 1.f / (1.f + expf(-v.w));
        const float gw = gr * cr + gg * cg + gb * cb + ga;
        const float dalpha = T * (gw - Q);
        Q = gw * alpha + (1.f - alpha + 1e-7f) * Q;
        const float ex = expf(x);
        const float dsp = x > 20.f ? 1.f : ex / (1.f + ex);           // F.softplus'(x), threshold 20
        const float w = alpha * T;
        const float4 d = make_float4(dalpha * (e * dist) * dsp, gr * w * cr * (1.f - cr), gg * w * cg * (1.f - cg), gb * w * cb * (1.f - cb));
        if (fromB) put(dvn, drow_n, idx, d); else put(dvc, drow_c, idx, d);
        znext = zcur;
    }
}

// Wave-per-ray version of the two compositing kernels for fitting batches (a few thousand rays: one thread per ray leaves the GPU
// empty and walks 2 x 256 dependent loads).  The merge of the two sorted depth lists is done by ranks (position of a coarse depth =
// its index + the number of new depths strictly below it; of a new depth = its index + the number of coarse depths <= it - the
// order k_composite's "za <= zb" walk produces), transmittance by a multiplicative wave scan, and the backward recurrence
// Q_{s-1} = gw_s alpha_s + (1 - alpha_s + 1e-7) Q_s by a suffix scan of affine maps.  Same formulas as the serial kernels; products
// and sums are associated differently (last-bit differences).  N, Ni <= 256.
constexpr int CW_MAX = 512;
template <bool BWD>
__global__ __launch_bounds__(256) void k_composite_wave(const CompBwdArgs b) {
    const CompArgs &a = b.c;
    __shared__ float s_za[4][CW_MAX / 2], s_zb[4][CW_MAX / 2], s_z[4][CW_MAX + 1], s_al[4][CW_MAX], s_T[4][CW_MAX];
    __shared__ int s_code[4][CW_MAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long tiles_n = (a.R + 31) / 32;
    const long long ray_raw = (long long)blockIdx.x * 4 + wv;
    const bool pad = ray_raw >= a.R;                       // padding rays of the last tile: zero deltas
    const long long ray = pad ? a.R - 1 : ray_raw;
    const bool live = ray_raw < tiles_n * 32;
    const long long tile = ray_raw >> 5;
    const int r = (int)(ray_raw & 31);
    const int N = a.N, Ni = a.Ni, S = N + Ni;
    float *za_ = s_za[wv], *zb_ = s_zb[wv], *zS = s_z[wv], *alS = s_al[wv], *TS = s_T[wv];
    int *codeS = s_code[wv];
    const long long tr = live ? tile : tiles_n - 1;
    const float4 *vc = a.vc + tr * 32 * (long long)N + r, *vn = a.vn + tr * 32 * (long long)Ni + r;
    const float *zn = a.zn + tr * 32 * (long long)Ni + r;
    const float nr = a.near[ray], fr = a.far[ray];
    for (int i = lane; i < N; i += 64) {
        float z;
        if (a.zc) z = a.zc[ray * N + i];
        else { const float t = linspace01(i, N); z = nr * (1.f - t) + fr * t; }
        za_[i] = z;
    }
    for (int j = lane; j < Ni; j += 64) zb_[j] = zn[32LL * j];
    __syncthreads();
    for (int i = lane; i < N; i += 64) {       // # new depths strictly below za[i]
        const float z = za_[i];
        int lo = 0, hi = Ni;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (zb_[mid] < z) lo = mid + 1; else hi = mid; }
        zS[i + lo] = z; codeS[i + lo] = i;
    }
    for (int j = lane; j < Ni; j += 64) {      // # coarse depths <= zb[j]
        const float z = zb_[j];
        int lo = 0, hi = N;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (za_[mid] <= z) lo = mid + 1; else hi = mid; }
        zS[j + lo] = z; codeS[j + lo] = j | (int)0x80000000;
    }
    __syncthreads();
    auto record = [&](int code) -> float4 { return code < 0 ? vn[32LL * (code & 0x7fffffff)] : vc[32LL * code]; };
    auto sraw = [&](const float4 &v, int s) -> float { return a.noise ? v.x + a.noise[ray * S + s] : v.x; };
    // forward: alpha, transmittance, and (forward kernel) the composited outputs
    float carry = 1.f, acc_w = 0.f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        float alpha = 0.f, zc = 0.f;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < S) {
            zc = zS[s];
            v = record(codeS[s]);
            const float dist = (s + 1 < S) ? zS[s + 1] - zc : 1e10f;
            alpha = 1.f - expf(-softplus_exact(sraw(v, s)) * dist);
            alS[s] = alpha;
        }
        const float fct = (s < S) ? (1.f - alpha + 1e-7f) : 1.f;
        const float incl = wave_incl_scan_mul(fct, lane);
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        if (s < S) TS[s] = T;
        carry *= __shfl(incl, 63);
        if constexpr (!BWD) {
            const float w = alpha * T;
            acc_w += w;
            acc_r += (1.f / (1.f + expf(-v.y))) * w;
            acc_g += (1.f / (1.f + expf(-v.z))) * w;
            acc_b += (1.f / (1.f + expf(-v.w))) * w;
            acc_d += w * zc;
        }
    }
    if constexpr (!BWD) {
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) {
            acc_w += __shfl_xor(acc_w, dd); acc_r += __shfl_xor(acc_r, dd); acc_g += __shfl_xor(acc_g, dd);
            acc_b += __shfl_xor(acc_b, dd); acc_d += __shfl_xor(acc_d, dd);
        }
        if (lane == 0 && !pad) {
            if (a.flags & HL_RENDER_WHITE_BKGD) { const float bg = 1.f - acc_w; acc_r += bg; acc_g += bg; acc_b += bg; }
            if (a.flags & HL_RENDER_NORMALIZE_DEPTH) {
                acc_d = (acc_d - nr) / (fr - nr + 1e-5f);
                if (a.flags & HL_RENDER_CLAMP_DEPTH) {
                    acc_d = acc_d > 1.f ? 1.f : acc_d;
                    acc_d = acc_d < 0.f ? 0.f : acc_d;
                }
            }
            a.rgb[ray * 3 + 0] = acc_r; a.rgb[ray * 3 + 1] = acc_g; a.rgb[ray * 3 + 2] = acc_b;
            a.acc[ray] = acc_w; a.depth[ray] = acc_d;
        }
    } else {
        if (!live) return;
        float4 *dvc = b.dvc + tile * 32 * (long long)N + r, *dvn = b.dvn + tile * 32 * (long long)Ni + r;
        float *drow_c = b.del + (long long)DROW_REC * b.del_stride + tile * 32 * (long long)N + r;
        float *drow_n = b.del + (long long)DROW_REC * b.del_stride + tiles_n * 32 * (long long)N + tile * 32 * (long long)Ni + r;
        const float gr = b.g_rgb[ray * 3 + 0], gg = b.g_rgb[ray * 3 + 1], gb = b.g_rgb[ray * 3 + 2];
        const float ga = b.g_acc[ray] - ((a.flags & HL_RENDER_WHITE_BKGD) ? gr + gg + gb : 0.f);   // rgb += 1 - acc
        float Qc = 0.f;                                        // Q just past the chunk being processed
        for (int base = ((S - 1) / 64) * 64; base >= 0; base -= 64) {
            const int s = base + lane;
            const bool on = s < S;
            float alpha = 0.f, T = 0.f, zc = 0.f, x = 0.f, dist = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, gw = 0.f;
            int code = 0;
            if (on) {
                code = codeS[s];
                const float4 v = record(code);
                alpha = alS[s]; T = TS[s]; zc = zS[s];
                dist = (s + 1 < S) ? zS[s + 1] - zc : 1e10f;
                x = sraw(v, s);
                cr = 1.f / (1.f + expf(-v.y)); cg = 1.f / (1.f + expf(-v.z)); cb = 1.f / (1.f + expf(-v.w));
                gw = gr * cr + gg * cg + gb * cb + ga;
            }
            // inclusive suffix scan of the maps M_s(Q) = bq + fq Q over the lanes: I_s = M_s o M_{s+1} o ... o M_{base+63}
            float fq = on ? (1.f - alpha + 1e-7f) : 1.f, bq = on ? gw * alpha : 0.f;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const float f2 = __shfl_down(fq, dd), b2 = __shfl_down(bq, dd);
                if (lane + dd < 64) { bq = bq + fq * b2; fq = fq * f2; }
            }
            // Q_s = I_{s+1}(Qc); the last lane sees Qc itself
            float fn = __shfl_down(fq, 1), bn = __shfl_down(bq, 1);
            if (lane == 63) { fn = 1.f; bn = 0.f; }
            const float Q = bn + fn * Qc;
            Qc = __shfl(bq, 0) + __shfl(fq, 0) * Qc;
            if (on) {
                float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!pad) {
                    const float e = expf(-softplus_exact(x) * dist);
                    const float dalpha = T * (gw - Q);
                    const float ex = expf(x);
                    const float dsp = x > 20.f ? 1.f : ex / (1.f + ex);           // F.softplus'(x), threshold 20
                    const float w = alpha * T;
                    d = make_float4(dalpha * (e * dist) * dsp, gr * w * cr * (1.f - cr), gg * w * cg * (1.f - cg), gb * w * cb * (1.f - cb));
                }
                const int idx = code & 0x7fffffff;
                float4 *rec = code < 0 ? dvn : dvc;
                float *row = code < 0 ? drow_n : drow_c;
                rec[32LL * idx] = d;
                row[32LL * idx] = d.x;
                row[32LL * idx + b.del_stride] = d.y;
                row[32LL * idx + 2 * b.del_stride] = d.z;
                row[32LL * idx + 3 * b.del_stride] = d.w;
            }
        }
    }
}

// Transposed weights for the backward-data products, in the same 16 KB chunk geometry the forward ring uses
// ([tile t][step/4][lane 64][4 steps]): delta_in[out] = sum_u W[u][col0 + out] delta_out[u], with the delta_out units u taken in
// accumulator-register order (unit_of), so the deltas - like the activations in the forward pass - go from one layer's accumulators
// straight into the next layer's B operand.
constexpr int NCH_BWD = 16;
__constant__ ChunkDesc c_chunks_bwd[NCH_BWD] = {
    {4, 155, 0, 3, 0, 4, 16},  {4, 155, 0, 3, 16, 4, 16},                                                       // views^T (feature columns)
    {3, 128, 0, 3, 0, 4, 16},  {3, 128, 0, 3, 16, 4, 16}, {3, 128, 0, 3, 32, 4, 16}, {3, 128, 0, 3, 48, 4, 16},  // feature^T
    {2, 155, 0, 4, 0, 1, 64},                                                                                   // pts2^T, tri-plane feature columns
    {2, 155, 27, 3, 0, 4, 16}, {2, 155, 27, 3, 16, 4, 16}, {2, 155, 27, 3, 32, 4, 16}, {2, 155, 27, 3, 48, 4, 16},  // pts2^T, hidden columns
    {1, 128, 0, 3, 0, 4, 16},  {1, 128, 0, 3, 16, 4, 16}, {1, 128, 0, 3, 32, 4, 16}, {1, 128, 0, 3, 48, 4, 16},  // pts1^T
    {0, 27, 0, 4, 0, 1, 64},                                                                                    // pts0^T
};

__global__ void k_pack_mlp_bwd(PackArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NCH_BWD * CHUNK_FLOATS) return;
    const int c = idx / CHUNK_FLOATS, e = idx % CHUNK_FLOATS;
    const ChunkDesc d = c_chunks_bwd[c];
    const int ns4 = d.nsteps / 4;
    const int k4 = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
    const int s4 = rest % ns4, t = rest / ns4;
    float v = 0.f;
    if (t < d.nt) {
        const int sp = d.base + s4 * 4 + k4, half = lane >> 5, out = 32 * t + (lane & 31);
        const int u = unit_of(sp >> 4, sp & 15, half);
        if (d.kind == 3 || out < 27) v = a.w[d.w][u * d.ld + d.col0 + out];
    }
    a.out[idx] = v;
}

// The same transposed weights as two fp16 planes (nearest even at both levels) for k_mlp_bwd<., true>: the u32x4 at [(t * ns4 + s4) * 64 + lane] of a chunk holds, for s4 even,
// plane 0 of the eight contraction steps 4 s4 .. 4 s4 + 7 of this lane half, for s4 odd plane 1 of the steps 4 (s4 - 1) .. - the byte count and the ring of the fp32 image.
__global__ void k_pack_mlp_bwd_h2(PackArgs a, unsigned short *out, const float *__restrict__ wsc) {   // wsc[l]: the power of two layer l's planes are multiplied by (k_mlp_scales_h2)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NCH_BWD * CHUNK_FLOATS * 2) return;
    const int c = idx / (CHUNK_FLOATS * 2), e = idx % (CHUNK_FLOATS * 2);
    const ChunkDesc d = c_chunks_bwd[c];
    const int ns4 = d.nsteps / 4;
    const int q = e & 7, lane = (e >> 3) & 63, rest = e >> 9;
    const int s4 = rest % ns4, t = rest / ns4;
    float v = 0.f;
    if (t < d.nt) {
        const int sp = d.base + (s4 >> 1) * 8 + q, half = lane >> 5, o = 32 * t + (lane & 31);
        const int u = unit_of(sp >> 4, sp & 15, half);
        if (d.kind == 3 || o < 27) v = a.w[d.w][u * d.ld + d.col0 + o] * wsc[d.w];
    }
    _Float16 h = (_Float16)v;
    if (s4 & 1) h = (_Float16)(v - (float)h);
    out[idx] = __builtin_bit_cast(unsigned short, h);
}

// fp16x2 form of mma16 (k_mlp_bwd<., true>): bp[jj][plane] = the B tile's registers 8 jj .. 8 jj + 7 of this lane half as two fp16 planes (scaled by the caller);
// the weights' planes from the LDS chunk (k_pack_mlp_bwd_h2); three partial products per k-step, smallest first
typedef _Float16 bwd_h8 __attribute__((ext_vector_type(8)));
template <int NT, int NS4>
__device__ __forceinline__ void mma16_h2(f32x16 (&acc)[NT], const u32x4 (&bp)[2][2], const f32x4 *__restrict__ ldsA, int s4base, int lane) {
    const u32x4 *A = reinterpret_cast<const u32x4 *>(ldsA);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        u32x4 a0[NT], a1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { a0[t] = A[(t * NS4 + s4base + 2 * jj) * 64 + lane]; a1[t] = A[(t * NS4 + s4base + 2 * jj + 1) * 64 + lane]; }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bwd_h8, a1[t]), __builtin_bit_cast(bwd_h8, bp[jj][0]), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bwd_h8, a0[t]), __builtin_bit_cast(bwd_h8, bp[jj][1]), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bwd_h8, a0[t]), __builtin_bit_cast(bwd_h8, bp[jj][0]), acc[t], 0, 0, 0);
        }
    }
}
// power-of-two scale that brings the largest |value| of a column's delta tiles to [2^13, 2^14) (deltas are far below fp16's normal range), and its inverse
template <int NT>
__device__ __forceinline__ void bwd_scale(const f32x16 (&v)[NT], int ntiles, float &sc, float &inv) {
    unsigned m = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < ntiles) {
#pragma unroll
            for (int r = 0; r < 16; ++r) m = max(m, __builtin_bit_cast(unsigned, v[t][r]) & 0x7fffffffu);
        }
    m = max(m, (unsigned)__shfl_xor((int)m, 32));          // (both lane halves of a column: one scale per ray and sample)
    const int eb = (int)(m >> 23);
    const bool ok = eb >= 40 && eb <= 240;                     // (zero / denormal / not finite: no scaling)
    sc = ok ? __builtin_bit_cast(float, (unsigned)(267 - eb) << 23) : 1.f;
    inv = ok ? __builtin_bit_cast(float, (unsigned)(eb - 13) << 23) : 1.f;
}
__device__ __forceinline__ void bwd_split(const f32x16 &v, float sc, u32x4 (&bp)[2][2]) {
    f32x16 w;
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = v[r] * sc;
    u32x4 p[2];
    split_h2t(w, 0, p); bp[0][0] = p[0]; bp[0][1] = p[1];
    split_h2t(w, 1, p); bp[1][0] = p[0]; bp[1][1] = p[1];
}

// One wave owns 32 rays as in k_march; per sample point: deltas of views_linear, feature_linear, pts_linears.2/1/0 by MFMA against
// the transposed weights (ring of 16 chunks), each multiplied by softplus'(pre) = 1 - exp(-activation) read back from the
// activation matrix, stored for the weight gradients; the two 27-wide feature deltas (skip connection + first layer) are summed
// in one accumulator tile and scattered to the tri-plane gradient through the four bilinear taps of each feature.
// H2 (round 5): the five transposed-weight products with fp16x2 operands on v_mfma_f32_32x32x16_f16 (weights: k_pack_mlp_bwd_h2's planes behind the fp32 image; the delta
// tiles split in registers, scaled per ray and sample by a power of two so that their largest entry sits in [2^13, 2^14) - deltas are far below fp16's normal range - and the
// products scaled back exactly): 384 MFMAs of 32 cycles per sample and wave instead of 1 024 of 64.
template <int NWV, bool H2 = false>
__global__ __launch_bounds__(NWV * 64, 2) void k_mlp_bwd(const MarchArgs a) {
#if __HIP_DEVICE_COMPILE__
    constexpr int NT = NWV * 64, NST = 1024 / NT;
    __shared__ __attribute__((aligned(16))) float lds[2 * CHUNK_FLOATS + SMALL_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const long long tile = (long long)blockIdx.x * NWV + (tid >> 6);
    const long long ray = tile * 32 + (lane & 31);
    const long long rc = ray < a.R ? ray : a.R - 1;
    const long long tiles_n = (a.R + 31) / 32;
    const bool tile_on = tile < tiles_n;
    const long long zt_base = (tile < tiles_n ? tile : tiles_n - 1) * 32 * (long long)a.S + (lane & 31);

    f32x4 *ldsv = reinterpret_cast<f32x4 *>(lds);
    const float *small = lds + 2 * CHUNK_FLOATS;
    const f32x4 *gsmall = reinterpret_cast<const f32x4 *>(a.packed) + NCH_FULL * CHUNK_FLOATS / 4;
    const f32x4 *gw = reinterpret_cast<const f32x4 *>(a.bwd_packed) + (H2 ? NCH_BWD * CHUNK_FLOATS / 4 : 0);
    const float *bsc = a.bwd_packed + 2 * NCH_BWD * CHUNK_FLOATS;   // H2: [l] the power of two layer l's planes carry, [8 + l] its inverse (k_mlp_scales_h2)
    for (int i = tid; i < SMALL_FLOATS / 4; i += NT) ldsv[2 * CHUNK_FLOATS / 4 + i] = gsmall[i];
#pragma unroll
    for (int q = 0; q < 2 * NST; ++q) ldsv[q * NT + tid] = gw[q * NT + tid];
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)gw, (short)0, NCH_BWD * CHUNK_FLOATS * 4, 0x00020000);
    const int wv = tid * 16;
    auto ldw = [&](int f4_index) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wv, f4_index * 16, 0));
    };
    f32x4 st[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) st[q] = ldw(2048 + q * NT);
    int cur = 0;

    const float ox = a.rays_o[rc * 3 + 0], oy = a.rays_o[rc * 3 + 1], oz = a.rays_o[rc * 3 + 2];
    const float dx = a.rays_d[rc * 3 + 0], dy = a.rays_d[rc * 3 + 1], dz = a.rays_d[rc * 3 + 2];
    const float nr = a.near[rc], fr = a.far[rc];
    const int S = a.S;
    const float offH = (float)(1.0 / (double)a.H);
    const float bmin0 = a.bounds[0], bmin1 = a.bounds[1], bmin2 = a.bounds[2];
    const float bext0 = a.bounds[3] - bmin0, bext1 = a.bounds[4] - bmin1, bext2 = a.bounds[5] - bmin2;
    __syncthreads();

#define HL_BWD_ADVANCE(g)                                                                                        \
    __syncthreads();                                                                                             \
    cur ^= 1024;                                                                                                 \
    _Pragma("unroll") for (int q_ = 0; q_ < NST; ++q_) ldsv[(cur ^ 1024) + q_ * NT + tid] = st[q_];              \
    _Pragma("unroll") for (int q_ = 0; q_ < NST; ++q_) st[q_] = ldw((((g) + 2) % NCH_BWD) * 1024 + q_ * NT);

    // activation -> softplus'(pre-activation) = sigmoid(pre) = 1 - exp(-softplus(pre))
    const unsigned act_stride4 = (unsigned)a.act_stride * 4u, del_stride4 = (unsigned)a.del_stride * 4u;
    const __amdgpu_buffer_rsrc_t act_rd = __builtin_amdgcn_make_buffer_rsrc((void *)a.act, (short)0, (int)((unsigned)ACT_ROWS * act_stride4), 0x00020000);
    const i32x4 del_rs = matrix_rsrc(a.del, (unsigned)DEL_ROWS * del_stride4);
    // The activations a block of deltas is multiplied with are fetched one block AHEAD, into ACT: the hidden stores carry a "memory"
    // clobber, so the compiler cannot hoist a load above them itself, and a load issued where its value is needed waits out an HBM
    // round trip (the matrix was written a millisecond ago and is long out of the caches) five times per sample.
    auto fetch_act = [&](f32x16 (&dst)[4], unsigned acol, int row0, int ntiles) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < ntiles) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dst[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(act_rd, (int)acol, (int)((unsigned)(row0 + unit_of(t, r, 0)) * act_stride4), 0));
            }
    };
    auto dsp = [](float h) -> float { return 1.f - __expf(-h); };     // softplus'(pre) from softplus(pre)

    const int s_lo = (int)blockIdx.y * a.s_per, s_hi = min(S, s_lo + a.s_per);
    for (int s = s_lo; s < s_hi; ++s) {
        const long long col = zt_base + 32LL * s;
        const unsigned actp = (unsigned)(a.act_off + col) * 4u + (unsigned)half * 4u * act_stride4;
        const unsigned delp = (unsigned)(a.del_off + col) * 4u + (unsigned)half * 4u * del_stride4;
        const float4 d = a.d_rec[col];      // (dsigma, dr, dg, db); zero on padding rays
        float zc;
        if (a.z) zc = a.z_tiled ? a.z[col] : a.z[rc * S + s];
        else { const float t = linspace01(s, S); zc = nr * (1.f - t) + fr * t; }

        // ---- rgb_linear^T, softplus' of views_linear ----
        f32x16 G[4], D[4], AV[4], ACT[4];
        fetch_act(AV, actp, ROW_V, 2);
        fetch_act(ACT, actp, ROW_X2, 4);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = (t * 2 + half) * 16 + r;
                const float g = small[SM_RW + o] * d.y + small[SM_RW + 64 + o] * d.z + small[SM_RW + 128 + o] * d.w;
                G[t][r] = g * dsp(AV[t][r]);
            }
        if (tile_on) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) hidden_store(del_rs, delp, (unsigned)(DROW_V + unit_of(t, r, 0)) * del_stride4, G[t][r]);
        }
        // ---- views_linear^T -> delta of feature_linear's output (no activation) ----
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) D[t][r] = 0.f;
        float sc = 1.f, inv = 1.f;
        u32x4 bp[2][2];
        if constexpr (H2) {
            bwd_scale<4>(G, 2, sc, inv);
            inv *= bsc[8 + 4];                                       // (views_linear's planes carry bsc[4]: the products leave it with the delta scale)
            bwd_split(G[0], sc, bp);
            mma16_h2<4, 4>(D, bp, ldsv + cur, 0, lane);
            HL_BWD_ADVANCE(1)
            bwd_split(G[1], sc, bp);
            mma16_h2<4, 4>(D, bp, ldsv + cur, 0, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) D[t][r] *= inv;
        } else {
            mma16<4, 4, 16>(D, G[0], ldsv + cur, 0, lane);
            HL_BWD_ADVANCE(1)
            mma16<4, 4, 16>(D, G[1], ldsv + cur, 0, lane);
        }
        if (tile_on) store_rows<4>(del_rs, delp, del_stride4, DROW_Y, D);
        // ---- feature_linear^T + alpha_linear^T, softplus' of pts_linears.2 ----
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) G[t][r] = small[SM_AW + (t * 2 + half) * 16 + r] * d.x;
        if constexpr (H2) {
            bwd_scale<4>(D, 4, sc, inv);
            inv *= bsc[8 + 3];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) G[t][r] *= sc * bsc[3];  // (the alpha head's term rides in the scaled accumulator: powers of two, exact)
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            HL_BWD_ADVANCE(2 + k)
            if constexpr (H2) { bwd_split(D[k], sc, bp); mma16_h2<4, 4>(G, bp, ldsv + cur, 0, lane); }
            else mma16<4, 4, 16>(G, D[k], ldsv + cur, 0, lane);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) G[t][r] *= dsp(ACT[t][r]) * (H2 ? inv : 1.f);
        fetch_act(ACT, actp, ROW_X1, 4);
        if (tile_on) store_rows<4>(del_rs, delp, del_stride4, DROW_X2, G);
        // ---- pts_linears.2^T: tri-plane feature columns -> DF, hidden columns -> delta of pts_linears.1 ----
        f32x16 DF[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) DF[0][r] = 0.f;
        HL_BWD_ADVANCE(6)
        if constexpr (H2) { bwd_scale<4>(G, 4, sc, inv); inv *= bsc[8 + 2]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (H2) { bwd_split(G[k], sc, bp); mma16_h2<1, 16>(DF, bp, ldsv + cur, 4 * k, lane); }
            else mma16<1, 16, 16>(DF, G[k], ldsv + cur, 4 * k, lane);
        }
        if constexpr (H2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) DF[0][r] *= inv;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) D[t][r] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            HL_BWD_ADVANCE(7 + k)
            if constexpr (H2) { bwd_split(G[k], sc, bp); mma16_h2<4, 4>(D, bp, ldsv + cur, 0, lane); }
            else mma16<4, 4, 16>(D, G[k], ldsv + cur, 0, lane);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) D[t][r] *= dsp(ACT[t][r]) * (H2 ? inv : 1.f);
        fetch_act(ACT, actp, ROW_X0, 4);
        if (tile_on) store_rows<4>(del_rs, delp, del_stride4, DROW_X1, D);
        // ---- pts_linears.1^T, softplus' of pts_linears.0 ----
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) G[t][r] = 0.f;
        if constexpr (H2) { bwd_scale<4>(D, 4, sc, inv); inv *= bsc[8 + 1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            HL_BWD_ADVANCE(11 + k)
            if constexpr (H2) { bwd_split(D[k], sc, bp); mma16_h2<4, 4>(G, bp, ldsv + cur, 0, lane); }
            else mma16<4, 4, 16>(G, D[k], ldsv + cur, 0, lane);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) G[t][r] *= dsp(ACT[t][r]) * (H2 ? inv : 1.f);
        if (tile_on) store_rows<4>(del_rs, delp, del_stride4, DROW_X0, G);
        // ---- pts_linears.0^T -> DF ----
        HL_BWD_ADVANCE(15)
        if constexpr (H2) {
            bwd_scale<4>(G, 4, sc, inv);
            inv *= bsc[8 + 0];
#pragma unroll
            for (int r = 0; r < 16; ++r) DF[0][r] *= sc * bsc[0];     // (the skip connection's part, already there: scaled along, exact)
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (H2) { bwd_split(G[k], sc, bp); mma16_h2<1, 16>(DF, bp, ldsv + cur, 4 * k, lane); }
            else mma16<1, 16, 16>(DF, G[k], ldsv + cur, 4 * k, lane);
        }
        if constexpr (H2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) DF[0][r] *= inv;
        }
        HL_BWD_ADVANCE(0)   // chunk 0 of the next sample

        // ---- d/d(tri-plane features) -> rows DROW_DF.. (k_plane_scatter sends them through the bilinear taps) ----
        if (tile_on) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (unit_of(0, r, half) < 27) hidden_store(del_rs, delp, (unsigned)(DROW_DF + unit_of(0, r, 0)) * del_stride4, DF[0][r]);
        }
    }
#undef HL_BWD_ADVANCE
#endif
}

// The feature-delta rows come out of k_mlp_bwd in the record order (tile-major: 32 rays per sample side by side); the scatter kernels
// walk one ray per wave with lanes = samples, which in that order is a 4-byte gather per lane (measured 0.9-1.8 GB of sector traffic
// for 57 MB of payload).  k_df_transpose rewrites the 27 rows ray-major (row DROW_DFT + k, column pass offset + ray * S + s).
// Each workgroup also records the largest |value| it moved (as a bit pattern: NaN > inf > finite) in wmax[tile * 27 + k]: the scatter kernels
// derive the scale of their fixed-point accumulators from the maximum over all of them (sc_fixed_scale).
struct DfTransposeArgs {
    float *del;
    long long del_stride, R;
    int N, Ni;
    unsigned *wmax;
};
__global__ __launch_bounds__(256) void k_df_transpose(const DfTransposeArgs a) {
    __shared__ float t[64][33];
    __shared__ unsigned wred[4];
    unsigned vmax = 0;
    const int k = blockIdx.x;
    const long long tile = blockIdx.y;
    const long long tiles_n = (a.R + 31) / 32, colsA = tiles_n * 32 * a.N;
    const float *src = a.del + (long long)(DROW_DF + k) * a.del_stride;
    float *dst = a.del + (long long)(DROW_DFT + k) * a.del_stride;
    for (int pass = 0; pass < 2; ++pass) {
        const int S = pass ? a.Ni : a.N;
        const long long base = pass ? colsA : 0;
        for (int s0 = 0; s0 < S; s0 += 64) {
            for (int i = threadIdx.x; i < 64 * 32; i += 256) {
                const int s = s0 + (i >> 5), r = i & 31;
                if (s < S) { const float v = src[base + (tile * S + s) * 32 + r]; t[i >> 5][r] = v; vmax = max(vmax, __float_as_uint(v) & 0x7fffffffu); }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < 64 * 32; i += 256) {
                const int r = i >> 6, s = s0 + (i & 63);
                const long long ray = tile * 32 + r;
                if (s < S && ray < a.R) dst[base + ray * S + s] = t[i & 63][r];
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) vmax = max(vmax, (unsigned)__shfl_xor((int)vmax, d));
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) a.wmax[tile * 27 + k] = max(max(wred[0], wred[1]), max(wred[2], wred[3]));
}

// Fixed-point accumulation of the scatter kernels (bit-reproducible: integer additions commute).  vmax = the largest |feature delta| of the
// call (2^E <= vmax < 2^(E+1)), n_pts = sample points: a texel's sum of |delta x bilinear weight| stays below n_pts x 2^(E+1), so with the unit
// 2^-k, k = 61 - ceil(log2 n_pts) - (E + 1), the 64-bit accumulator cannot overflow and a contribution is rounded at 2^-(43 ... 62) of vmax -
// below fp32's own rounding of the sum.  Returns false when a delta is not finite (the outputs are NaN then).
__device__ __forceinline__ bool sc_fixed_scale(const unsigned *wmax, long long n_wmax, long long n_pts, unsigned *red, double &scale, double &inv) {
    unsigned m = 0;
    for (long long i = threadIdx.x; i < n_wmax; i += blockDim.x) m = max(m, wmax[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = max(m, red[w]);
    __syncthreads();
    if (m >= 0x7f800000u) return false;
    const int E = (int)(m >> 23) - 127;
    int cl = 0;
    while ((1LL << cl) < n_pts) ++cl;
    const int k = 61 - cl - (E + 1);
    scale = ldexp(1.0, k);
    inv = ldexp(1.0, -k);
    return true;
}
__device__ __forceinline__ void sc_add(unsigned long long *t, float c, double scale) {
    atomicAdd(t, (unsigned long long)__double2ll_rn((double)c * scale));
}

// Tri-plane gradient: transpose of the bilinear lookup [renderer.py:502-531].  Sending every sample point's 27 feature deltas
// through their four taps with global float atomics costs 108 atomics per point (measured: 1.1 ms per 262 144 points, 2/3 of the
// backward kernel, bound by the L2 atomic units).  Instead one workgroup OWNS a 16x16-texel tile of one (plane, group) image
// - 3 channels, 3 KB of LDS; 2 304 workgroups of 512 threads at 256x256: measured best of 8..64-texel tiles - scans all sample points of the batch (recomputing the forward pass's texel coordinates bit for bit),
// accumulates the taps that fall into its tile with LDS atomics, and writes the tile out with plain stores: no global atomics, no
// zero-fill; the LDS accumulators are 64-bit fixed point (sc_fixed_scale), so the result is the same bits on every run.
struct ScatterArgs {
    const float *rays_o, *rays_d, *near, *far, *bounds;
    const float *zc, *zn;        // coarse depths: rows (R,N) or null -> linspace; new depths: rows (R,Ni) (zn_rows) or tile-major
    int zn_rows;
    long long R;
    int N, Ni, H, W;
    const float *del;            // rows DROW_DF..+26, columns: coarse pass then new depths
    long long del_stride;
    float *dplanes;              // (27, H, W), overwritten
    const unsigned *wmax;        // k_df_transpose's per-workgroup maxima
};
constexpr int SC_TILE = 16, SC_THREADS = 512;

__global__ __launch_bounds__(SC_THREADS) void k_plane_scatter(const ScatterArgs a) {
    __shared__ unsigned long long acc[3 * SC_TILE * SC_TILE];
    __shared__ unsigned red[SC_THREADS / 64];
    const int q = blockIdx.x, p = q / 3, g = q % 3;
    const int tiles_x = (a.W + SC_TILE - 1) / SC_TILE;
    const int tx0 = (int)(blockIdx.y % tiles_x) * SC_TILE, ty0 = (int)(blockIdx.y / tiles_x) * SC_TILE;
    for (int i = threadIdx.x; i < 3 * SC_TILE * SC_TILE; i += SC_THREADS) acc[i] = 0ull;
    double scale, inv;
    const bool finite = sc_fixed_scale(a.wmax, ((a.R + 31) / 32) * 27, a.R * (long long)(a.N + a.Ni), red, scale, inv);      // (ends with a barrier)
    if (!finite) {
        for (int i = threadIdx.x; i < 3 * SC_TILE * SC_TILE; i += SC_THREADS) {
            const int c = i / (SC_TILE * SC_TILE), y = (i / SC_TILE) % SC_TILE + ty0, x = i % SC_TILE + tx0;
            if (y < a.H && x < a.W) a.dplanes[((long long)(3 * q + c) * a.H + y) * a.W + x] = __builtin_nanf("");
        }
        return;
    }
    const long long tiles_n = (a.R + 31) / 32;
    const long long colsA = tiles_n * 32 * a.N;
    const float offH = (float)(1.0 / (double)a.H);
    const float bmin0 = a.bounds[0], bmin1 = a.bounds[1], bmin2 = a.bounds[2];
    const float bext0 = a.bounds[3] - bmin0, bext1 = a.bounds[4] - bmin1, bext2 = a.bounds[5] - bmin2;
    const float *df = a.del + (long long)(DROW_DFT + 3 * q) * a.del_stride;      // ray-major copy (k_df_transpose)
    // texel coordinates of the point at depth z of a ray - the forward pass's arithmetic, bit for bit
    auto texel = [&](const float (&o)[3], const float (&d)[3], float z, float &ix, float &iy) {
        const float px = o[0] + d[0] * z, py = o[1] + d[1] * z, pz = o[2] + d[2] * z;
        const float nx = 2.f * (px - bmin0) / bext0 - 1.f;
        const float ny = 2.f * (py - bmin1) / bext1 - 1.f;
        const float nz = 2.f * (pz - bmin2) / bext2 - 1.f;
        float gu = (p == 2) ? nz : nx;
        float gv = (p == 1) ? nz : ny;
        gu = (g == 1) ? gu + offH : gu;
        gv = (g == 2) ? gv + offH : gv;
        ix = ((gu + 1.f) * (float)a.W - 1.f) / 2.f;
        iy = ((gv + 1.f) * (float)a.H - 1.f) / 2.f;
    };
    // One WAVE per ray, lanes = 64 consecutive samples.  The texel coordinates are linear in the depth, so the samples that can touch
    // this tile lie in one depth interval [za, zb] (computed with a margin; the exact test below decides): most rays miss the tile
    // (tested 64 rays at a time, one per lane), and along a ray that crosses it the lanes inside the interval are neighbours
    // (a thread-per-ray layout puts 64 unrelated rays in a wave and executes the tap code for nearly every sample of every ray).
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    auto bcast = [](float v, int j) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); };
    for (long long base = (long long)wv * 64; base < a.R; base += SC_THREADS) {
      // 64 rays at a time, one per lane: load them and intersect them with the tile in parallel, then visit the hits one by one
      const long long lray = base + lane < a.R ? base + lane : a.R - 1;
      const float lo_[3] = {a.rays_o[lray * 3 + 0], a.rays_o[lray * 3 + 1], a.rays_o[lray * 3 + 2]};
      const float ld_[3] = {a.rays_d[lray * 3 + 0], a.rays_d[lray * 3 + 1], a.rays_d[lray * 3 + 2]};
      const float lnr = a.near[lray], lfr = a.far[lray];
      float lza = -3.0e38f, lzb = 3.0e38f;
      {
        float x_a, y_a, x_b, y_b;
        texel(lo_, ld_, 0.f, x_a, y_a);
        texel(lo_, ld_, 1.f, x_b, y_b);
        auto clip = [&](float c0, float slope, float lo, float hi) {       // lo <= c0 + slope z < hi
            if (fabsf(slope) < 1e-9f) { if (c0 < lo || c0 >= hi) { lza = 1.f; lzb = 0.f; } return; }
            const float z1 = (lo - c0) / slope, z2 = (hi - c0) / slope;
            lza = fmaxf(lza, fminf(z1, z2));
            lzb = fminf(lzb, fmaxf(z1, z2));
        };
        clip(x_a, x_b - x_a, (float)tx0 - 1.05f, (float)(tx0 + SC_TILE) + 0.05f);
        clip(y_a, y_b - y_a, (float)ty0 - 1.05f, (float)(ty0 + SC_TILE) + 0.05f);
        lza -= 1e-4f * fabsf(lza) + 1e-6f;
        lzb += 1e-4f * fabsf(lzb) + 1e-6f;
      }
      unsigned long long hits = __ballot((lza <= lzb) && (base + lane < a.R));
      while (hits) {
        const int j = __builtin_ctzll(hits);
        hits &= hits - 1;
        const long long ray = base + j;
        const long long tile = ray >> 5;
        const int rl = (int)(ray & 31);
        const float o[3] = {bcast(lo_[0], j), bcast(lo_[1], j), bcast(lo_[2], j)};
        const float d[3] = {bcast(ld_[0], j), bcast(ld_[1], j), bcast(ld_[2], j)};
        const float nr = bcast(lnr, j), fr = bcast(lfr, j), za = bcast(lza, j), zb = bcast(lzb, j);
        for (int pass = 0; pass < 2; ++pass) {
            const int S = pass ? a.Ni : a.N;
            const long long col0 = (pass ? colsA : 0) + ray * S;
            for (int s0 = 0; s0 < S; s0 += 64) {
                const int si = s0 + lane;
                if (si >= S) continue;
                float z;
                if (pass) z = a.zn_rows ? a.zn[ray * a.Ni + si] : a.zn[tile * 32LL * S + rl + 32LL * si];
                else if (a.zc) z = a.zc[ray * a.N + si];
                else { const float t = linspace01(si, a.N); z = nr * (1.f - t) + fr * t; }
                if (z < za || z > zb) continue;
                const long long col = col0 + si;
                float ix, iy;
                texel(o, d, z, ix, iy);
                const float x0f = floorf(ix), y0f = floorf(iy);
                const float x1f = x0f + 1.f, y1f = y0f + 1.f;
                if (!(x0f >= (float)(tx0 - 1) && x0f < (float)(tx0 + SC_TILE) && y0f >= (float)(ty0 - 1) && y0f < (float)(ty0 + SC_TILE))) continue;
                const int x0 = (int)x0f - tx0, y0 = (int)y0f - ty0, x1 = x0 + 1, y1 = y0 + 1;     // tile coordinates
                // inside the tile (and, through the tile's extent, inside the image: taps outside contribute nothing - zeros padding)
                const bool vx0 = (x0 >= 0) & (x0 < SC_TILE) & (x0 + tx0 < a.W), vx1 = (x1 >= 0) & (x1 < SC_TILE) & (x1 + tx0 < a.W);
                const bool vy0 = (y0 >= 0) & (y0 < SC_TILE) & (y0 + ty0 < a.H), vy1 = (y1 >= 0) & (y1 < SC_TILE) & (y1 + ty0 < a.H);
                const float w_nw = (x1f - ix) * (y1f - iy), w_ne = (ix - x0f) * (y1f - iy);
                const float w_sw = (x1f - ix) * (iy - y0f), w_se = (ix - x0f) * (iy - y0f);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v = df[(long long)c * a.del_stride + col];
                    unsigned long long *t = acc + c * SC_TILE * SC_TILE;
                    if (vx0 & vy0) sc_add(t + y0 * SC_TILE + x0, v * w_nw, scale);
                    if (vx1 & vy0) sc_add(t + y0 * SC_TILE + x1, v * w_ne, scale);
                    if (vx0 & vy1) sc_add(t + y1 * SC_TILE + x0, v * w_sw, scale);
                    if (vx1 & vy1) sc_add(t + y1 * SC_TILE + x1, v * w_se, scale);
                }
            }
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * SC_TILE * SC_TILE; i += SC_THREADS) {
        const int c = i / (SC_TILE * SC_TILE), y = (i / SC_TILE) % SC_TILE + ty0, x = i % SC_TILE + tx0;
        if (y < a.H && x < a.W) a.dplanes[((long long)(3 * q + c) * a.H + y) * a.W + x] = (float)((double)(long long)acc[i] * inv);
    }
}

// Tri-plane gradient for sample points that are not on straight rays in tri-plane space (canonical-space training: the points went
// through the body deformation).  Same ownership scheme as k_plane_scatter; the culling unit is a block of 64 consecutive samples of
// one ray, whose bounding box in normalised plane coordinates k_block_bbox computes once: a workgroup tests 64 boxes per wave step
// against its tile and walks only the blocks that can touch it (lanes = the block's samples).
struct ScatterPtsArgs {
    const float4 *pts[2];        // canonical sample points of the two passes, tile-major [R/32][S][32]
    const float *bounds;         // (2,3) box of the canonical space (t_world_bounds)
    long long R;
    int N, Ni, H, W;
    const float *del;
    long long del_stride;
    float *dplanes;
    float *bbox;                 // [blocks][6]: min xyz, max xyz of the block's normalised coordinates
    const unsigned *wmax;        // k_df_transpose's per-workgroup maxima
};

__device__ __forceinline__ void scatter_blocks(const ScatterPtsArgs &a, int &nbA, int &nbB) { nbA = (a.N + 63) / 64; nbB = (a.Ni + 63) / 64; }

__global__ __launch_bounds__(256) void k_block_bbox(const ScatterPtsArgs a) {
    // one wave per (ray, block): block index = ray * (nbA + nbB) + b
    int nbA, nbB;
    scatter_blocks(a, nbA, nbB);
    const int lane = threadIdx.x & 63;
    const long long blk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= a.R * (nbA + nbB)) return;
    const long long ray = blk / (nbA + nbB);
    const int b = (int)(blk - ray * (nbA + nbB));
    const int pass = b >= nbA, S = pass ? a.Ni : a.N, s = 64 * (pass ? b - nbA : b) + lane;
    const float bmin0 = a.bounds[0], bmin1 = a.bounds[1], bmin2 = a.bounds[2];
    const float bext0 = a.bounds[3] - bmin0, bext1 = a.bounds[4] - bmin1, bext2 = a.bounds[5] - bmin2;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (s < S) {
        const float4 p = a.pts[pass][((ray >> 5) * S + s) * 32 + (ray & 31)];
        lo[0] = hi[0] = 2.f * (p.x - bmin0) / bext0 - 1.f;
        lo[1] = hi[1] = 2.f * (p.y - bmin1) / bext1 - 1.f;
        lo[2] = hi[2] = 2.f * (p.z - bmin2) / bext2 - 1.f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int d = 32; d > 0; d >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], d)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d)); }
    if (lane < 6) a.bbox[blk * 6 + lane] = lane < 3 ? lo[lane] : hi[lane - 3];
}

__global__ __launch_bounds__(SC_THREADS) void k_plane_scatter_pts(const ScatterPtsArgs a) {
    __shared__ unsigned long long acc[3 * SC_TILE * SC_TILE];
    __shared__ unsigned red[SC_THREADS / 64];
    const int q = blockIdx.x, p = q / 3, g = q % 3;
    const int tiles_x = (a.W + SC_TILE - 1) / SC_TILE;
    const int tx0 = (int)(blockIdx.y % tiles_x) * SC_TILE, ty0 = (int)(blockIdx.y / tiles_x) * SC_TILE;
    for (int i = threadIdx.x; i < 3 * SC_TILE * SC_TILE; i += SC_THREADS) acc[i] = 0ull;
    double scale, inv;
    const bool finite = sc_fixed_scale(a.wmax, ((a.R + 31) / 32) * 27, a.R * (long long)(a.N + a.Ni), red, scale, inv);      // (ends with a barrier)
    if (!finite) {
        for (int i = threadIdx.x; i < 3 * SC_TILE * SC_TILE; i += SC_THREADS) {
            const int c = i / (SC_TILE * SC_TILE), y = (i / SC_TILE) % SC_TILE + ty0, x = i % SC_TILE + tx0;
            if (y < a.H && x < a.W) a.dplanes[((long long)(3 * q + c) * a.H + y) * a.W + x] = __builtin_nanf("");
        }
        return;
    }
    int nbA, nbB;
    scatter_blocks(a, nbA, nbB);
    const long long tiles_n = (a.R + 31) / 32;
    const long long colsA = tiles_n * 32 * a.N;
    const long long nblk = a.R * (nbA + nbB);
    const float offH = (float)(1.0 / (double)a.H);
    const float bmin0 = a.bounds[0], bmin1 = a.bounds[1], bmin2 = a.bounds[2];
    const float bext0 = a.bounds[3] - bmin0, bext1 = a.bounds[4] - bmin1, bext2 = a.bounds[5] - bmin2;
    const float *df = a.del + (long long)(DROW_DFT + 3 * q) * a.del_stride;      // ray-major copy (k_df_transpose)
    const int cu = (p == 2) ? 2 : 0, cv = (p == 1) ? 2 : 1;          // which normalised coordinate drives u / v of this plane
    auto to_px = [](float gn, int size) { return ((gn + 1.f) * (float)size - 1.f) / 2.f; };
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (long long base = (long long)wv * 64; base < nblk; base += SC_THREADS) {
        const long long lb = base + lane < nblk ? base + lane : nblk - 1;
        const float *bb = a.bbox + lb * 6;
        // texel range the block can touch (monotone maps; the group shift only moves it up by 1/H): [floor(lo), floor(hi) + 1]
        const float ulo = to_px(bb[cu], a.W) - 0.01f, uhi = to_px(bb[3 + cu] + offH, a.W) + 1.01f;
        const float vlo = to_px(bb[cv], a.H) - 0.01f, vhi = to_px(bb[3 + cv] + offH, a.H) + 1.01f;
        const bool hit = base + lane < nblk && uhi >= (float)tx0 && ulo < (float)(tx0 + SC_TILE) && vhi >= (float)ty0 && vlo < (float)(ty0 + SC_TILE);
        unsigned long long hits = __ballot(hit);
        while (hits) {
            const int j = __builtin_ctzll(hits);
            hits &= hits - 1;
            const long long blk = base + j;
            const long long ray = blk / (nbA + nbB);
            const int b = (int)(blk - ray * (nbA + nbB));
            const int pass = b >= nbA, S = pass ? a.Ni : a.N, s = 64 * (pass ? b - nbA : b) + lane;
            if (s >= S) continue;
            const long long lc = ((ray >> 5) * S + s) * 32 + (ray & 31);
            const float4 pt = a.pts[pass][lc];
            const long long col = (pass ? colsA : 0) + ray * S + s;
            const float nx = 2.f * (pt.x - bmin0) / bext0 - 1.f;
            const float ny = 2.f * (pt.y - bmin1) / bext1 - 1.f;
            const float nz = 2.f * (pt.z - bmin2) / bext2 - 1.f;
            float gu = (p == 2) ? nz : nx;
            float gv = (p == 1) ? nz : ny;
            gu = (g == 1) ? gu + offH : gu;
            gv = (g == 2) ? gv + offH : gv;
            const float ix = ((gu + 1.f) * (float)a.W - 1.f) / 2.f;
            const float iy = ((gv + 1.f) * (float)a.H - 1.f) / 2.f;
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float x1f = x0f + 1.f, y1f = y0f + 1.f;
            if (!(x0f >= (float)(tx0 - 1) && x0f < (float)(tx0 + SC_TILE) && y0f >= (float)(ty0 - 1) && y0f < (float)(ty0 + SC_TILE))) continue;
            const int x0 = (int)x0f - tx0, y0 = (int)y0f - ty0, x1 = x0 + 1, y1 = y0 + 1;
            const bool vx0 = (x0 >= 0) & (x0 < SC_TILE) & (x0 + tx0 < a.W), vx1 = (x1 >= 0) & (x1 < SC_TILE) & (x1 + tx0 < a.W);
            const bool vy0 = (y0 >= 0) & (y0 < SC_TILE) & (y0 + ty0 < a.H), vy1 = (y1 >= 0) & (y1 < SC_TILE) & (y1 + ty0 < a.H);
            const float w_nw = (x1f - ix) * (y1f - iy), w_ne = (ix - x0f) * (y1f - iy);
            const float w_sw = (x1f - ix) * (iy - y0f), w_se = (ix - x0f) * (iy - y0f);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = df[(long long)c * a.del_stride + col];
                unsigned long long *t = acc + c * SC_TILE * SC_TILE;
                if (vx0 & vy0) sc_add(t + y0 * SC_TILE + x0, v * w_nw, scale);
                if (vx1 & vy0) sc_add(t + y0 * SC_TILE + x1, v * w_ne, scale);
                if (vx0 & vy1) sc_add(t + y1 * SC_TILE + x0, v * w_sw, scale);
                if (vx1 & vy1) sc_add(t + y1 * SC_TILE + x1, v * w_se, scale);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * SC_TILE * SC_TILE; i += SC_THREADS) {
        const int c = i / (SC_TILE * SC_TILE), y = (i / SC_TILE) % SC_TILE + ty0, x = i % SC_TILE + tx0;
        if (y < a.H && x < a.W) a.dplanes[((long long)(3 * q + c) * a.H + y) * a.W + x] = (float)((double)(long long)acc[i] * inv);
    }
}

// Weight gradients: C[m][n] += sum_p delta[a_row0 + m][p] * act[b_row0 + n][p] over the sample points - both operands are rows of
// the two matrices, contiguous along the reduction.  A workgroup takes one layer and a range of points; per 32 points it stages the
// layer's delta rows (<= 128) and activation rows (<= 155) through LDS with full-line loads (8 lanes x 16 B per row), then wave w
// multiplies delta rows 32w..32w+31 against all activation rows: a lane reads 16 points of "its" row per operand tile (lanes 0-31:
// points 0-15, lanes 32-63: points 16-31; row pitch 36 floats keeps the 16-byte reads conflict-free) = two 8-wide k-groups of
// v_mfma_f32_32x32x16_bf16.  The fp32 products come from exact three-way bf16 splits of both operands (truncation: v_and / v_perm /
// v_sub), six partial products per k-group, fp32 accumulation - 12 MFMAs of 32 cycles where v_mfma_f32_32x32x2_f32 needs 16 of 64
// (round 4: 0.91 -> 0.68 ms per subject, 3.1 -> 4.2 TB/s; the fp32 MFMAs, not HBM, were what the kernel waited for).  The row sums of
// delta (the bias gradients) fall out of the A operand on the VALU.  Partial results of the point ranges are added to the gradient
// tensors with float atomics.
struct WgradJob {
    int a_row0, M, b_row0, N;   // rows of the A / B operand; C[i][n] goes to out[i * ld_i + n * ld_n]
    int a_is_act;               // 0: A rows from the delta matrix, B rows from the activation matrix; 1: the other way round (the two
                                // heads, 1 and 3 delta rows: as A they would occupy a whole 32-row slab of one wave per workgroup)
    int ld_i, ld_n;
    float *out, *bias_out;      // bias = row sums of the delta operand
};
constexpr int WGRAD_JOBS = 7, WGRAD_MAX_NB = 5, WG_PITCH = 36, WG_AROWS = 128, WG_ROWS = WG_AROWS + 32 * WGRAD_MAX_NB, WG_LD = WG_ROWS / 32;
struct WgradArgs {
    WgradJob job[WGRAD_JOBS];
    const float *del, *act;
    long long del_stride, act_stride, n_cols;
    int k_per_wg;                          // points per workgroup (multiple of 32)
    float *part;                           // partial results [point range][part_len]: job j at part_off[j], C[i][n] at i * N + n, the bias sums behind
    int part_off[WGRAD_JOBS], part_len;
};

// eight consecutive points of a row -> the three bf16 planes of one v_mfma_f32_32x32x16_bf16 operand (exact split by truncation)
__device__ __forceinline__ void wg_split8(const f32x4 &lo, const f32x4 &hi4, u32x4 (&pl)[3]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x = q < 2 ? lo[2 * q] : hi4[2 * q - 4], y = q < 2 ? lo[2 * q + 1] : hi4[2 * q - 3];
        pl[0][q] = b3_pack(x, y);
        x -= __builtin_bit_cast(float, b3_hi(x)); y -= __builtin_bit_cast(float, b3_hi(y));
        pl[1][q] = b3_pack(x, y);
        x -= __builtin_bit_cast(float, b3_hi(x)); y -= __builtin_bit_cast(float, b3_hi(y));
        pl[2][q] = b3_pack(x, y);
    }
}

__global__ __launch_bounds__(256, 2) void k_wgrad(const WgradArgs a) {
#if __HIP_DEVICE_COMPILE__
    __shared__ __attribute__((aligned(16))) float lds[WG_ROWS * WG_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, row = lane & 31;
    const WgradJob jb = a.job[blockIdx.x];
    const long long p_lo = (long long)blockIdx.y * a.k_per_wg;
    if (p_lo >= a.n_cols) return;
    const long long p_hi = p_lo + a.k_per_wg < a.n_cols ? p_lo + a.k_per_wg : a.n_cols;
    const unsigned ds4 = (unsigned)a.del_stride * 4u, as4 = (unsigned)a.act_stride * 4u;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)a.del, (short)0, (int)((unsigned)DEL_ROWS * ds4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)a.act, (short)0, (int)((unsigned)ACT_ROWS * as4), 0x00020000);
    const unsigned OOB = 0x80000000u;   // rows past M / N read zeros
    const int NB = (jb.N + 31) / 32, MT = (jb.M + 31) / 32;
    const unsigned a_s4 = jb.a_is_act ? as4 : ds4, b_s4 = jb.a_is_act ? ds4 : as4;
    // staging: thread = (row group tid >> 3, 16-byte piece tid & 7); pass i covers staged rows 32 i .. 32 i + 31
    const int piece = tid & 7, rg = tid >> 3;
    unsigned goff[WG_LD];
#pragma unroll
    for (int i = 0; i < WG_LD; ++i) {
        const int r = 32 * i + rg;
        if (i < WG_AROWS / 32) goff[i] = r < jb.M ? (unsigned)(jb.a_row0 + r) * a_s4 + 16u * piece : OOB;
        else goff[i] = (r - WG_AROWS) < jb.N ? (unsigned)(jb.b_row0 + r - WG_AROWS) * b_s4 + 16u * piece : OOB;
    }
    f32x4 st[WG_LD];
    auto fetch = [&](long long p) {
        const int p4 = (int)((unsigned)p * 4u);
#pragma unroll
        for (int i = 0; i < WG_LD; ++i) {
            const bool used = i < WG_AROWS / 32 ? i < MT : (i - WG_AROWS / 32) < NB;
            if (used) st[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((i < WG_AROWS / 32) != (jb.a_is_act != 0) ? rd : ra, (int)goff[i], p4, 0));
        }
    };
    f32x16 acc[WGRAD_MAX_NB];
#pragma unroll
    for (int t = 0; t < WGRAD_MAX_NB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    f32x4 *ldsv = reinterpret_cast<f32x4 *>(lds);
    fetch(p_lo);
    for (long long p = p_lo; p < p_hi; p += 32) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WG_LD; ++i) {
            const bool used = i < WG_AROWS / 32 ? i < MT : (i - WG_AROWS / 32) < NB;
            if (used) ldsv[(32 * i + rg) * (WG_PITCH / 4) + piece] = st[i];
        }
        __syncthreads();
        if (p + 32 < p_hi) fetch(p + 32);
        if (wave < MT) {
            f32x4 av[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = ldsv[(32 * wave + row) * (WG_PITCH / 4) + 4 * half + q];
            {
                // fp32 products from exact three-way bf16 splits of both operands (six partial products, smallest first, fp32 accumulation):
                // a lane's 16 points are two 8-wide k-groups of v_mfma_f32_32x32x16_bf16; 12 MFMAs of 32 cycles replace 16 of 64
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
                u32x4 ap[2][3];
                wg_split8(av[0], av[1], ap[0]);
                wg_split8(av[2], av[3], ap[1]);
#pragma unroll
                for (int t = 0; t < WGRAD_MAX_NB; ++t) {
                    if (t < NB) {
                        f32x4 bv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) bv[q] = ldsv[(WG_AROWS + 32 * t + row) * (WG_PITCH / 4) + 4 * half + q];
                        u32x4 bp[2][3];
                        wg_split8(bv[0], bv[1], bp[0]);
                        wg_split8(bv[2], bv[3], bp[1]);
#pragma unroll
                        for (int i = 0; i < 6; ++i)
#pragma unroll
                            for (int b = 0; b < 2; ++b)
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[b][PA[i]]), __builtin_bit_cast(bf16x8, bp[b][PB[i]]), acc[t], 0, 0, 0);
                    }
                }
            }
            if (!jb.a_is_act) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bsum += (av[q][0] + av[q][1]) + (av[q][2] + av[q][3]);
            } else if (wave == 0) {      // delta rows are the B operand (one column tile): their sums from this wave's B fragment
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bv = ldsv[(WG_AROWS + row) * (WG_PITCH / 4) + 4 * half + q];
                    bsum += (bv[0] + bv[1]) + (bv[2] + bv[3]);
                }
            }
        }
    }
    if (wave >= MT) return;
    float *pt = a.part + (long long)blockIdx.y * a.part_len + a.part_off[blockIdx.x];
    // C[i][n]: lane holds column n = 32 t + row and rows i = 32 wave + unit_of(0, r, half)
#pragma unroll
    for (int t = 0; t < WGRAD_MAX_NB; ++t) {
        const int n = 32 * t + row;
        if (t < NB && n < jb.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 32 * wave + unit_of(0, r, half);
                if (i < jb.M) pt[i * jb.N + n] = acc[t][r];
            }
        }
    }
    bsum += __shfl_xor(bsum, 32);
    const int m = 32 * wave + row;
    if (!jb.a_is_act) { if (half == 0 && m < jb.M) pt[jb.M * jb.N + m] = bsum; }
    else if (wave == 0 && half == 0 && row < jb.N) pt[jb.M * jb.N + row] = bsum;
#endif
}

// the point ranges' partial results summed in a FIXED order and added to the gradient tensors (plain stores: the result is the same bits on every run)
__global__ __launch_bounds__(256) void k_wgrad_finish(const WgradArgs a, int n_ranges) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= a.part_len) return;
    int j = 0;
#pragma unroll
    for (int q = 1; q < WGRAD_JOBS; ++q) j = e >= a.part_off[q] ? q : j;
    const WgradJob jb = a.job[j];
    const int le = e - a.part_off[j];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four interleaved chains (ranges y = 0, 4, 8 ... / 1, 5, ... ), always combined the same way
    const float *src = a.part + e;
    int y = 0;
    for (; y + 4 <= n_ranges; y += 4) {
        s0 += src[(long long)y * a.part_len];
        s1 += src[(long long)(y + 1) * a.part_len];
        s2 += src[(long long)(y + 2) * a.part_len];
        s3 += src[(long long)(y + 3) * a.part_len];
    }
    for (; y < n_ranges; ++y) s0 += src[(long long)y * a.part_len];
    const float tot = (s0 + s1) + (s2 + s3);
    if (le < jb.M * jb.N) {
        const int i = le / jb.N, n = le - i * jb.N;
        jb.out[(long long)i * jb.ld_i + (long long)n * jb.ld_n] += tot;
    } else {
        jb.bias_out[le - jb.M * jb.N] += tot;
    }
}

extern "C" {

// fp32 image + the fp16 fragments of k_march16 + the three bf16 planes of k_march_b3
// (+ the two fp16 planes of the fp16x2 products)
size_t hl_render_mlp_packed_bytes(void) { return (size_t)PACKED_FLOATS * sizeof(float) + (size_t)P16_FRAGS * 1024 + B3_BYTES + (size_t)P16_FRAGS * 2 * 1024; }

int hl_render_mlp_pack(const hl_render_mlp_params *p, void *packed, void *stream) {
    HL_REQUIRE(p && packed, "hl_render_mlp_pack: null argument");
    PackArgs a;
    a.w[0] = p->pts0_w; a.w[1] = p->pts1_w; a.w[2] = p->pts2_w; a.w[3] = p->feat_w; a.w[4] = p->views_w;
    a.b[0] = p->pts0_b; a.b[1] = p->pts1_b; a.b[2] = p->pts2_b; a.b[3] = p->feat_b; a.b[4] = p->views_b;
    a.alpha_w = p->alpha_w; a.alpha_b = p->alpha_b; a.rgb_w = p->rgb_w; a.rgb_b = p->rgb_b;
    for (int i = 0; i < 5; ++i) HL_REQUIRE(a.w[i] && a.b[i], "hl_render_mlp_pack: null weight %d", i);
    HL_REQUIRE(a.alpha_w && a.alpha_b && a.rgb_w && a.rgb_b, "hl_render_mlp_pack: null head weight");
    a.out = (float *)packed;
    hipLaunchKernelGGL(k_pack_mlp, dim3((PACKED_FLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    int rc = hl::check_launch("k_pack_mlp");
    if (rc) return rc;
    hipLaunchKernelGGL(k_pack_mlp16, dim3(P16_FRAGS * 2), dim3(256), 0, (hipStream_t)stream, a,
                       reinterpret_cast<unsigned short *>(static_cast<float *>(packed) + PACKED_FLOATS));
    rc = hl::check_launch("k_pack_mlp16");
    if (rc) return rc;
    hipLaunchKernelGGL(k_pack_mlp_b3, dim3(P16_FRAGS * 3 * 2), dim3(256), 0, (hipStream_t)stream, a,
                       reinterpret_cast<unsigned short *>(static_cast<char *>(packed) + (size_t)PACKED_FLOATS * sizeof(float) + (size_t)P16_FRAGS * 1024));
    rc = hl::check_launch("k_pack_mlp_b3");
    if (rc) return rc;
    hipLaunchKernelGGL(k_mlp_scales_h2, dim3(5), dim3(256), 0, (hipStream_t)stream, a, a.out + NCH_FULL * CHUNK_FLOATS + SM_SC);      // (behind k_pack_mlp, which wrote 1s there; in front of the planes that read them)
    rc = hl::check_launch("k_mlp_scales_h2");
    if (rc) return rc;
    hipLaunchKernelGGL(k_pack_mlp_h2, dim3(P16_FRAGS * 2 * 2), dim3(256), 0, (hipStream_t)stream, a,
                       reinterpret_cast<unsigned short *>(static_cast<char *>(packed) + (size_t)PACKED_FLOATS * sizeof(float) + (size_t)P16_FRAGS * 1024 + B3_BYTES));
    return hl::check_launch("k_pack_mlp_h2");
}

size_t hl_planes_packed_bytes(int H, int W) { return (size_t)9 * H * W * sizeof(float4); }

int hl_planes_pack(const float *planes, int H, int W, void *packed, void *stream) {
    HL_REQUIRE(planes && packed && H > 0 && W > 0, "hl_planes_pack: bad argument");
    const int64_t n = (int64_t)9 * H * W;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_pack_planes, dim3(grid), dim3(256), 0, (hipStream_t)stream, planes, (float4 *)packed, H, W);
    return hl::check_launch("k_pack_planes");
}

static inline int64_t tiles32(int64_t n_rays) { return (n_rays + 31) / 32; }
// k_importance: a wave walks 8 rays of its tile one after the other; ray batches that would launch fewer than ~1024 workgroups that way
// (fitting: 2048 rays = 64 tiles) give each wave fewer rays and the tile more workgroups
static inline int imp_rays_per_wave(int64_t n_rays) {
    int rpw = 8;
    while (rpw > 1 && tiles32(n_rays) * (8 / rpw) < 1024) rpw >>= 1;
    return rpw;
}

size_t hl_render_workspace_bytes(int64_t n_rays, int n_samples, int n_importance) {
    if (n_rays <= 0 || n_importance <= 0) return 256;
    // tile-major [ceil(R/32)][samples][32].  Evaluate-once pipeline: raw sample records (float4) of the n_samples coarse and
    // the n_importance new points + the sorted new depths; HL_RENDER_REEVALUATE: sigma (n_samples) + z_all (both) - smaller
    return (size_t)tiles32(n_rays) * 32 * ((size_t)(n_samples + n_importance) * 4 + n_importance) * sizeof(float) + 256;
}

static int fill_march(MarchArgs &a, const void *mlp, const void *planes, int H, int W, const float *bounds,
                      const float *ro, const float *rd, const float *nr, const float *fr) {
    HL_REQUIRE(mlp && planes && bounds && ro && rd && nr && fr, "render: null argument");
    HL_REQUIRE(H > 0 && W > 0, "render: bad plane size %dx%d", H, W);
    a.packed = (const float *)mlp;
    a.planes = (const float4 *)planes;
    a.H = H; a.W = W;
    a.bounds = bounds;
    a.rays_o = ro; a.rays_d = rd; a.near = nr; a.far = fr;
    return HL_OK;
}

int hl_render_coarse(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                     const float *rays_o, const float *rays_d, const float *near, const float *far,
                     const float *z_vals, int64_t n_rays, int n_samples, float *sigma_out, void *stream) {
    HL_REQUIRE(n_rays > 0 && n_samples >= 2 && sigma_out, "hl_render_coarse: bad argument");
    MarchArgs a{};
    int rcode = fill_march(a, mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far);
    if (rcode) return rcode;
    a.z = z_vals; a.z_tiled = 0; a.R = n_rays; a.S = n_samples; a.flags = 0; a.sigma_out = sigma_out;
    const unsigned grid = (unsigned)((n_rays + 255) / 256);
    hipLaunchKernelGGL(k_march<false>, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_march<coarse>");
}

int hl_render_importance(const float *sigma, const float *rays_d, const float *near, const float *far,
                         const float *z_vals, const float *u, int64_t n_rays, int n_samples, int n_importance,
                         float *z_all_out, void *stream) {
    HL_REQUIRE(sigma && rays_d && near && far && u && z_all_out, "hl_render_importance: null argument");
    HL_REQUIRE(n_rays > 0 && n_samples >= 3 && n_importance >= 1, "hl_render_importance: bad sizes");
    if (n_samples > IMP_MAX_N || n_importance > IMP_MAX_N)
        return hl::fail(HL_ERR_UNSUPPORTED, "hl_render_importance: n_samples/n_importance > %d", IMP_MAX_N);
    ImpArgs a{sigma, rays_d, near, far, z_vals, u, n_rays, n_samples, n_importance, z_all_out, 1, 0, imp_rays_per_wave(n_rays)};
    hipLaunchKernelGGL(k_importance, dim3((unsigned)(tiles32(n_rays) * (8 / a.rays_per_wave))), dim3(256), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_importance");
}

static int render_fine_impl(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                            const float *rays_o, const float *rays_d, const float *near, const float *far, const float *z_all,
                            int z_tiled, int64_t n_rays, int n_total_samples, unsigned flags, float *rgb, float *acc,
                            float *depth, void *stream) {
    HL_REQUIRE(n_rays > 0 && n_total_samples >= 2 && rgb && acc && depth, "hl_render_fine: bad argument");
    MarchArgs a{};
    int rcode = fill_march(a, mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far);
    if (rcode) return rcode;
    a.z = z_all; a.z_tiled = z_tiled; a.R = n_rays; a.S = n_total_samples; a.flags = flags;
    a.rgb = rgb; a.acc = acc; a.depth = depth;
    const unsigned grid = (unsigned)((n_rays + 255) / 256);
    hipLaunchKernelGGL(k_march<true>, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_march<fine>");
}

int hl_render_fine(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                   const float *rays_o, const float *rays_d, const float *near, const float *far, const float *z_all,
                   int z_tiled, int64_t n_rays, int n_total_samples, unsigned flags, float *rgb, float *acc, float *depth,
                   void *stream) {
    return render_fine_impl(mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far, z_all, z_tiled, n_rays,
                            n_total_samples, flags, rgb, acc, depth, stream);
}

static int render_eval_impl(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *rays_o,
                            const float *rays_d, const float *near, const float *far, const float *z, int z_tiled, int64_t n_rays,
                            int n_samples, float *records_out, int mlp_mode, void *stream) {   // mlp_mode: 0 fp32 MFMA, 1 fp16 operands (opt-in), 2 bf16x3 split products
    HL_REQUIRE(n_rays > 0 && n_samples >= 1 && records_out, "hl_render_eval: bad argument");
    MarchArgs a{};
    int rcode = fill_march(a, mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far);
    if (rcode) return rcode;
    a.z = z; a.z_tiled = z_tiled; a.R = n_rays; a.S = n_samples; a.flags = 0; a.vals_out = (float4 *)records_out;
    if (mlp_mode == 3) {   // HL_RENDER_MLP_FP16X2: two fp16 planes per operand, three partial products, fp32 accumulation
        const unsigned short *ph2 = reinterpret_cast<const unsigned short *>(static_cast<const char *>(mlp_packed) + (size_t)PACKED_FLOATS * sizeof(float) + (size_t)P16_FRAGS * 1024 + B3_BYTES);
        static const bool okh = hipFuncSetAttribute((const void *)k_march_plw<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PLW_LDS<2>) == hipSuccess;
        HL_REQUIRE(okh, "k_march_plw<2>: cannot raise the dynamic LDS limit to %zu bytes", PLW_LDS<2>);
        hipLaunchKernelGGL(k_march_plw<2>, dim3((unsigned)((n_rays + 255) / 256)), dim3(512), PLW_LDS<2>, (hipStream_t)stream, a, ph2);
        return hl::check_launch("k_march_plw<2>");
    }
    if (mlp_mode == 2) {   // HL_RENDER_MLP_BF16X3: exact three-way bf16 split of both operands, six partial products, fp32 accumulation
        const unsigned short *pb3 = reinterpret_cast<const unsigned short *>(static_cast<const char *>(mlp_packed) + (size_t)PACKED_FLOATS * sizeof(float) + (size_t)P16_FRAGS * 1024);
        static const int b3_waves = getenv("HL_B3_WAVES") ? atoi(getenv("HL_B3_WAVES")) : 8;   // developer switch: 4 = k_march_b3 (one wave per SIMD, hand-scheduled)
        if (b3_waves == 8) {
            static const bool okw = hipFuncSetAttribute((const void *)k_march_plw<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3W_LDS) == hipSuccess;
            HL_REQUIRE(okw, "k_march_b3w: cannot raise the dynamic LDS limit to %zu bytes", B3W_LDS);
            hipLaunchKernelGGL(k_march_plw<3>, dim3((unsigned)((n_rays + 255) / 256)), dim3(512), B3W_LDS, (hipStream_t)stream, a, pb3);
            return hl::check_launch("k_march_b3w");
        }
        const dim3 grid((unsigned)((n_rays + 127) / 128));
#define HL_B3_LAUNCH(A)                                                                                                                    \
    case A: {                                                                                                                              \
        static const bool ok_ = hipFuncSetAttribute((const void *)k_march_b3<A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3R_LDS) == hipSuccess; \
        HL_REQUIRE(ok_, "k_march_b3: cannot raise the dynamic LDS limit to %zu bytes", B3R_LDS);                                           \
        hipLaunchKernelGGL(k_march_b3<A>, grid, dim3(256), B3R_LDS, (hipStream_t)stream, a, pb3);                                          \
        break;                                                                                                                             \
    }
#ifdef HL_B3_ABLATIONS   // developer builds (HL_RENDER_FLAGS=-DHL_B3_ABLATIONS): timing ablations selected by the environment, wrong images
        static const int abl = getenv("HL_B3_ABL") ? atoi(getenv("HL_B3_ABL")) : 0;
        switch (abl) {
            HL_B3_LAUNCH(4) HL_B3_LAUNCH(64) HL_B3_LAUNCH(128) HL_B3_LAUNCH(256) HL_B3_LAUNCH(512) HL_B3_LAUNCH(1024) HL_B3_LAUNCH(2048)
            default: HL_B3_LAUNCH(0)
        }
#else
        switch (0) { default: HL_B3_LAUNCH(0) }
#endif
#undef HL_B3_LAUNCH
        return hl::check_launch("k_march_b3");
    }
    if (mlp_mode == 1) {   // HL_RENDER_MLP_FP16 (opt-in): fp16 operands, all weights LDS-resident
        const size_t sh = (size_t)P16_FRAGS * 1024 + SMALL_FLOATS * sizeof(float);
        static const bool attr_ok = hipFuncSetAttribute((const void *)k_march16<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)((size_t)P16_FRAGS * 1024 + SMALL_FLOATS * sizeof(float))) == hipSuccess &&
                                    hipFuncSetAttribute((const void *)k_march16<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)((size_t)P16_FRAGS * 1024 + SMALL_FLOATS * sizeof(float))) == hipSuccess;
        HL_REQUIRE(attr_ok, "k_march16: cannot raise the dynamic LDS limit to %zu bytes", sh);
        const unsigned short *p16 = reinterpret_cast<const unsigned short *>(static_cast<const float *>(mlp_packed) + PACKED_FLOATS);
        static const int nwv = getenv("HL_MARCH16_WAVES") ? atoi(getenv("HL_MARCH16_WAVES")) : 4;      // 4 = one wave per SIMD (512 registers, no spills: 20.1 against 21.9 ms per view); 8: developer switch
        if (nwv == 4) hipLaunchKernelGGL(k_march16<4>, dim3((unsigned)((n_rays + 127) / 128)), dim3(256), sh, (hipStream_t)stream, a, p16);
        else hipLaunchKernelGGL(k_march16<8>, dim3((unsigned)((n_rays + 255) / 256)), dim3(512), sh, (hipStream_t)stream, a, p16);
        return hl::check_launch("k_march16");
    }
    // (4-wave workgroups, two per CU with independent barriers, measured equal - 74.5 vs 74.7 ms per view - at twice the weight
    //  traffic: 8 waves it is)
    hipLaunchKernelGGL((k_march<true, true, 8>), dim3((unsigned)((n_rays + 255) / 256)), dim3(512), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_march<eval>");
}

// the one-pass fine launch (k_march_plw<NPL, false, true>): importance depths from the coarse records `vc`, their evaluation, merge + compositing - images out
static int render_onepass_fine(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *rays_o, const float *rays_d,
                               const float *near, const float *far, const float *vc, const float *u, float *zn_scratch, int64_t n_rays, int n_samples,
                               unsigned flags, int npl, float *rgb, float *acc, float *depth, void *stream) {
    HL_REQUIRE(vc && u && zn_scratch && rgb && acc && depth && n_samples >= 3 && n_samples <= 128, "render_onepass_fine: bad argument");
    MarchArgs a{};
    int rcode = fill_march(a, mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far);
    if (rcode) return rcode;
    a.z = zn_scratch; a.z_tiled = 1; a.R = n_rays; a.S = n_samples; a.flags = flags;
    if (const char *e_ = getenv("HL_ONEPASS_DEV")) a.flags |= (unsigned)strtoul(e_, nullptr, 0);
    a.fz_vc = (const float4 *)vc; a.fz_u = u; a.fz_zn = zn_scratch; a.fz_N = n_samples;
    a.rgb = rgb; a.acc = acc; a.depth = depth;
    const dim3 grid((unsigned)((n_rays + 255) / 256));
    if (npl == 2) {
        const unsigned short *ph2 = reinterpret_cast<const unsigned short *>(static_cast<const char *>(mlp_packed) + (size_t)PACKED_FLOATS * sizeof(float) + (size_t)P16_FRAGS * 1024 + B3_BYTES);
        static const bool okh = hipFuncSetAttribute((const void *)k_march_plw<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PLW_LDS_FUSE<2>) == hipSuccess;
        HL_REQUIRE(okh, "k_march_plw<2, one-pass>: cannot raise the dynamic LDS limit to %zu bytes", PLW_LDS_FUSE<2>);
        hipLaunchKernelGGL((k_march_plw<2, false, true>), grid, dim3(512), PLW_LDS_FUSE<2>, (hipStream_t)stream, a, ph2);
        return hl::check_launch("k_march_plw<2, one-pass>");
    }
    return hl::fail(HL_ERR_UNSUPPORTED, "render_onepass_fine: the one-pass launch exists for the fp16x2 products only");
}

int hl_render_eval(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *rays_o,
                   const float *rays_d, const float *near, const float *far, const float *z, int z_tiled, int64_t n_rays,
                   int n_samples, float *records_out, void *stream) {
    return render_eval_impl(mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far, z, z_tiled, n_rays, n_samples, records_out, 0,
                            stream);
}

int hl_render_eval_products(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *rays_o,
                            const float *rays_d, const float *near, const float *far, const float *z, int z_tiled, int64_t n_rays,
                            int n_samples, unsigned flags, float *records_out, void *stream) {
    return render_eval_impl(mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far, z, z_tiled, n_rays, n_samples, records_out,
                            (flags & HL_RENDER_MLP_FP16) ? 1 : ((flags & HL_RENDER_MLP_FP16X2) ? 3 : ((flags & HL_RENDER_MLP_BF16X3) ? 2 : 0)), stream);
}

int hl_render_importance_new(const float *records, const float *rays_d, const float *near, const float *far, const float *z_vals,
                             const float *u, int64_t n_rays, int n_samples, int n_importance, float *z_new_out, void *stream) {
    HL_REQUIRE(records && rays_d && near && far && u && z_new_out, "hl_render_importance_new: null argument");
    HL_REQUIRE(n_rays > 0 && n_samples >= 3 && n_importance >= 1, "hl_render_importance_new: bad sizes");
    if (n_samples > IMP_MAX_N || n_importance > IMP_MAX_N)
        return hl::fail(HL_ERR_UNSUPPORTED, "hl_render_importance_new: n_samples/n_importance > %d", IMP_MAX_N);
    ImpArgs a{records, rays_d, near, far, z_vals, u, n_rays, n_samples, n_importance, z_new_out, 4, 1, imp_rays_per_wave(n_rays)};
    hipLaunchKernelGGL(k_importance, dim3((unsigned)(tiles32(n_rays) * (8 / a.rays_per_wave))), dim3(256), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_importance<new>");
}

int hl_render_composite(const float *near, const float *far, const float *z_vals, const float *z_new, const float *rec_coarse,
                        const float *rec_new, int64_t n_rays, int n_samples, int n_importance, unsigned flags, float *rgb,
                        float *acc, float *depth, void *stream) {
    return hl_render_composite_noise(near, far, z_vals, z_new, rec_coarse, rec_new, nullptr, n_rays, n_samples, n_importance, flags, rgb,
                                     acc, depth, stream);
}

int hl_render_composite_noise(const float *near, const float *far, const float *z_vals, const float *z_new, const float *rec_coarse,
                              const float *rec_new, const float *noise, int64_t n_rays, int n_samples, int n_importance, unsigned flags,
                              float *rgb, float *acc, float *depth, void *stream) {
    HL_REQUIRE(near && far && z_new && rec_coarse && rec_new && rgb && acc && depth, "hl_render_composite: null argument");
    HL_REQUIRE(n_rays > 0 && n_samples >= 1 && n_importance >= 1, "hl_render_composite: bad sizes");
    CompArgs c{near, far, z_vals, z_new, (const float4 *)rec_coarse, (const float4 *)rec_new, n_rays, n_samples, n_importance,
               flags, rgb, acc, depth, noise};
    // fitting batches (training noise given, few rays): one wave per ray
    if (noise && n_rays <= 65536 && n_samples <= CW_MAX / 2 && n_importance <= CW_MAX / 2) {
        CompBwdArgs b{};
        b.c = c;
        hipLaunchKernelGGL(k_composite_wave<false>, dim3((unsigned)(tiles32(n_rays) * 8)), dim3(256), 0, (hipStream_t)stream, b);
        return hl::check_launch("k_composite_wave<fwd>");
    }
    // 80 KB of unused dynamic LDS per workgroup = two workgroups (eight waves) per CU: the merge walk of a wave keeps ~12 KB of record lines
    // open, and with 16+ waves per CU the lines are evicted before the neighbouring rays have used them (rocprofv3, 512x512 view: 976 us
    // unlimited, 954 us at four workgroups, 780 us at two, 1 150 us at one)
    constexpr int COMP_OCC_LDS = 80000;
    static const bool occ_ok = hipFuncSetAttribute((const void *)k_composite, hipFuncAttributeMaxDynamicSharedMemorySize, COMP_OCC_LDS) == hipSuccess;
    hipLaunchKernelGGL(k_composite, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), occ_ok ? COMP_OCC_LDS : 0, (hipStream_t)stream, c);
    return hl::check_launch("k_composite");
}

// ---- training: backward of the evaluate-once pipeline ----
// one workgroup per CU is resident (8 waves of ~240 VGPRs): aim at two rounds of 256 workgroups
static inline unsigned sample_splits(unsigned ray_groups, int n_samples) {
    unsigned s = (512 + ray_groups - 1) / ray_groups;
    if (s > (unsigned)n_samples) s = (unsigned)n_samples;
    return s < 1 ? 1 : s;
}
size_t hl_render_mlp_bwd_packed_bytes(void) { return (size_t)(2 * NCH_BWD * CHUNK_FLOATS + 16) * sizeof(float); }   // the fp32 image, the two fp16 planes (k_pack_mlp_bwd_h2), the planes' layer scales + inverses

int hl_render_mlp_pack_bwd(const hl_render_mlp_params *p, void *packed, void *stream) {
    HL_REQUIRE(p && packed, "hl_render_mlp_pack_bwd: null argument");
    PackArgs a{};
    a.w[0] = p->pts0_w; a.w[1] = p->pts1_w; a.w[2] = p->pts2_w; a.w[3] = p->feat_w; a.w[4] = p->views_w;
    for (int i = 0; i < 5; ++i) HL_REQUIRE(a.w[i], "hl_render_mlp_pack_bwd: null weight %d", i);
    a.out = (float *)packed;
    hipLaunchKernelGGL(k_pack_mlp_bwd, dim3((NCH_BWD * CHUNK_FLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    int rc_ = hl::check_launch("k_pack_mlp_bwd");
    if (rc_) return rc_;
    float *wsc = static_cast<float *>(packed) + 2 * NCH_BWD * CHUNK_FLOATS;
    hipLaunchKernelGGL(k_mlp_scales_h2, dim3(5), dim3(256), 0, (hipStream_t)stream, a, wsc);
    rc_ = hl::check_launch("k_mlp_scales_h2");
    if (rc_) return rc_;
    hipLaunchKernelGGL(k_pack_mlp_bwd_h2, dim3((NCH_BWD * CHUNK_FLOATS * 2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, a,
                       reinterpret_cast<unsigned short *>(static_cast<float *>(packed) + NCH_BWD * CHUNK_FLOATS), wsc);
    return hl::check_launch("k_pack_mlp_bwd_h2");
}

void hl_render_train_rows(int *act_rows, int *del_rows) {
    if (act_rows) *act_rows = ACT_ROWS;
    if (del_rows) *del_rows = DEL_ROWS;
}

int hl_render_eval_acts(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *rays_o,
                        const float *rays_d, const float *near, const float *far, const float *z, int z_tiled, int64_t n_rays,
                        int n_samples, float *records_out, float *act, int64_t act_stride, int64_t act_off, void *stream) {
    HL_REQUIRE(n_rays > 0 && n_samples >= 1 && records_out && act, "hl_render_eval_acts: bad argument");
    HL_REQUIRE(act_off >= 0 && act_stride >= act_off + tiles32(n_rays) * 32 * n_samples, "hl_render_eval_acts: activation rows too short");
    HL_REQUIRE((int64_t)ACT_ROWS * act_stride * 4 < (1LL << 32), "hl_render_eval_acts: activation matrix must stay below 4 GiB (row stride %lld)",
               (long long)act_stride);
    MarchArgs a{};
    int rcode = fill_march(a, mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far);
    if (rcode) return rcode;
    a.z = z; a.z_tiled = z_tiled; a.R = n_rays; a.S = n_samples; a.flags = 0; a.vals_out = (float4 *)records_out;
    a.act = act; a.act_stride = act_stride; a.act_off = act_off;
    const unsigned groups = (unsigned)((n_rays + 255) / 256);
    const unsigned splits = sample_splits(groups, n_samples);
    a.s_per = (n_samples + (int)splits - 1) / (int)splits;
    // round 5: the evaluate pass of the fitting step on the fp16x2 kernel (k_march_plw<2, ACTS>: the inference default's arithmetic; the backward reads the
    // activations it writes).  HL_FIT_FP32=1 (developer knob, read once): the fp32-MFMA kernel of rounds 1-4.
    static const int fit_fp32 = [] { const char *e_ = getenv("HL_FIT_FP32"); return e_ ? atoi(e_) : 0; }();
    const dim3 grid(groups, (unsigned)((n_samples + a.s_per - 1) / a.s_per));
    if (!fit_fp32) {
        const unsigned short *ph2 = reinterpret_cast<const unsigned short *>(static_cast<const char *>(mlp_packed) + (size_t)PACKED_FLOATS * sizeof(float) + (size_t)P16_FRAGS * 1024 + B3_BYTES);
        static const bool okh = hipFuncSetAttribute((const void *)k_march_plw<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PLW_LDS<2>) == hipSuccess;
        HL_REQUIRE(okh, "k_march_plw<2, acts>: cannot raise the dynamic LDS limit to %zu bytes", PLW_LDS<2>);
        hipLaunchKernelGGL((k_march_plw<2, true>), grid, dim3(512), PLW_LDS<2>, (hipStream_t)stream, a, ph2);
        return hl::check_launch("k_march_plw<2, acts>");
    }
    hipLaunchKernelGGL((k_march<true, true, 8, false, true>), grid, dim3(512), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_march<eval, acts>");
}

size_t hl_render_composite_backward_scratch_bytes(int64_t n_rays, int n_samples, int n_importance) {
    return (size_t)tiles32(n_rays) * 32 * (size_t)(n_samples + n_importance) * 8;
}

int hl_render_composite_backward(const float *near, const float *far, const float *z_vals, const float *z_new, const float *rec_coarse,
                                 const float *rec_new, const float *noise, const float *g_rgb, const float *g_acc, int64_t n_rays,
                                 int n_samples, int n_importance, unsigned flags, float *d_rec_coarse, float *d_rec_new, float *del,
                                 int64_t del_stride, void *scratch, void *stream) {
    HL_REQUIRE(near && far && z_new && rec_coarse && rec_new && g_rgb && g_acc && d_rec_coarse && d_rec_new && del && scratch,
               "hl_render_composite_backward: null argument");
    HL_REQUIRE(del_stride >= tiles32(n_rays) * 32 * (int64_t)(n_samples + n_importance), "hl_render_composite_backward: delta rows too short");
    HL_REQUIRE(n_rays > 0 && n_samples >= 1 && n_importance >= 1, "hl_render_composite_backward: bad sizes");
    CompBwdArgs b{};
    b.c = CompArgs{near, far, z_vals, z_new, (const float4 *)rec_coarse, (const float4 *)rec_new, n_rays, n_samples, n_importance,
                   flags, nullptr, nullptr, nullptr, noise};
    b.g_rgb = g_rgb; b.g_acc = g_acc;
    b.dvc = (float4 *)d_rec_coarse; b.dvn = (float4 *)d_rec_new;
    b.sT = (float *)scratch;
    b.sSrc = (int *)scratch + tiles32(n_rays) * 32 * (size_t)(n_samples + n_importance);
    b.del = del; b.del_stride = del_stride;
    if (n_samples <= CW_MAX / 2 && n_importance <= CW_MAX / 2) {
        hipLaunchKernelGGL(k_composite_wave<true>, dim3((unsigned)(tiles32(n_rays) * 8)), dim3(256), 0, (hipStream_t)stream, b);
        return hl::check_launch("k_composite_wave<bwd>");
    }
    hipLaunchKernelGGL(k_composite_bwd, dim3((unsigned)((tiles32(n_rays) * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b);
    return hl::check_launch("k_composite_bwd");
}

int hl_render_mlp_backward(const void *mlp_packed, const void *mlp_bwd_packed, int H, int W, const float *bounds, const float *rays_o,
                           const float *rays_d, const float *near, const float *far, const float *z, int z_tiled, int64_t n_rays,
                           int n_samples, const float *d_records, const float *act, int64_t act_stride, int64_t act_off, float *del,
                           int64_t del_stride, int64_t del_off, void *stream) {
    HL_REQUIRE(mlp_packed && mlp_bwd_packed && bounds && rays_o && rays_d && near && far && d_records && act && del,
               "hl_render_mlp_backward: null argument");
    HL_REQUIRE(n_rays > 0 && n_samples >= 1 && H > 0 && W > 0, "hl_render_mlp_backward: bad sizes");
    const int64_t cols = tiles32(n_rays) * 32 * n_samples;
    HL_REQUIRE(act_off >= 0 && act_stride >= act_off + cols && del_off >= 0 && del_stride >= del_off + cols,
               "hl_render_mlp_backward: activation / delta rows too short");
    HL_REQUIRE((int64_t)ACT_ROWS * act_stride * 4 < (1LL << 32) && (int64_t)DEL_ROWS * del_stride * 4 < (1LL << 32),
               "hl_render_mlp_backward: activation / delta matrices must stay below 4 GiB");
    MarchArgs a{};
    a.packed = (const float *)mlp_packed; a.bwd_packed = (const float *)mlp_bwd_packed;
    a.H = H; a.W = W; a.bounds = bounds;
    a.rays_o = rays_o; a.rays_d = rays_d; a.near = near; a.far = far;
    a.z = z; a.z_tiled = z_tiled; a.R = n_rays; a.S = n_samples;
    a.d_rec = (const float4 *)d_records;
    a.act = const_cast<float *>(act); a.act_stride = act_stride; a.act_off = act_off;
    a.del = del; a.del_stride = del_stride; a.del_off = del_off;
    const unsigned groups = (unsigned)((n_rays + 255) / 256);
    const unsigned splits = sample_splits(groups, n_samples);
    a.s_per = (n_samples + (int)splits - 1) / (int)splits;
    static const int fit_fp32 = [] { const char *e_ = getenv("HL_FIT_FP32"); return e_ ? atoi(e_) : 0; }();   // developer knob (read once): the fp32-MFMA kernel of rounds 1-4
    if (!fit_fp32) hipLaunchKernelGGL((k_mlp_bwd<8, true>), dim3(groups, (unsigned)((n_samples + a.s_per - 1) / a.s_per)), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_mlp_bwd<8, false>), dim3(groups, (unsigned)((n_samples + a.s_per - 1) / a.s_per)), dim3(512), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_mlp_bwd");
}

int hl_render_plane_grads(int H, int W, const float *bounds, const float *rays_o, const float *rays_d, const float *near, const float *far,
                          const float *z_vals, const float *z_new, int z_new_rows, int64_t n_rays, int n_samples, int n_importance,
                          const float *del, int64_t del_stride, float *d_planes, void *scratch, void *stream) {
    HL_REQUIRE(bounds && rays_o && rays_d && near && far && z_new && del && d_planes && scratch, "hl_render_plane_grads: null argument");
    HL_REQUIRE(H > 0 && W > 0 && n_rays > 0 && n_samples >= 1 && n_importance >= 1, "hl_render_plane_grads: bad sizes");
    HL_REQUIRE(del_stride >= tiles32(n_rays) * 32 * (int64_t)(n_samples + n_importance), "hl_render_plane_grads: delta rows too short");
    DfTransposeArgs tr{const_cast<float *>(del), del_stride, n_rays, n_samples, n_importance, (unsigned *)scratch};
    hipLaunchKernelGGL(k_df_transpose, dim3(27, (unsigned)tiles32(n_rays)), dim3(256), 0, (hipStream_t)stream, tr);
    ScatterArgs a{rays_o, rays_d, near, far, bounds, z_vals, z_new, z_new_rows, n_rays, n_samples, n_importance, H, W, del, del_stride, d_planes,
                  (const unsigned *)scratch};
    const unsigned tiles = (unsigned)(((W + SC_TILE - 1) / SC_TILE) * ((H + SC_TILE - 1) / SC_TILE));
    hipLaunchKernelGGL(k_plane_scatter, dim3(9, tiles), dim3(SC_THREADS), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_plane_scatter");
}

int hl_render_eval_points_acts(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *pts_c,
                               const float *dirs_c, int64_t n_rays, int n_samples, float *records_out, float *act, int64_t act_stride,
                               int64_t act_off, void *stream) {
    HL_REQUIRE(mlp_packed && planes_packed && bounds && pts_c && dirs_c && records_out && act, "hl_render_eval_points_acts: null argument");
    HL_REQUIRE(n_rays > 0 && n_samples >= 1 && H > 0 && W > 0, "hl_render_eval_points_acts: bad sizes");
    HL_REQUIRE(act_off >= 0 && act_stride >= act_off + tiles32(n_rays) * 32 * n_samples, "hl_render_eval_points_acts: activation rows too short");
    HL_REQUIRE((int64_t)ACT_ROWS * act_stride * 4 < (1LL << 32), "hl_render_eval_points_acts: activation matrix must stay below 4 GiB");
    MarchArgs a{};
    a.packed = (const float *)mlp_packed; a.planes = (const float4 *)planes_packed; a.H = H; a.W = W; a.bounds = bounds;
    a.rays_o = pts_c; a.rays_d = pts_c; a.near = pts_c; a.far = pts_c;      // placeholders, as in hl_render_eval_points
    a.z = nullptr; a.z_tiled = 0; a.R = n_rays; a.S = n_samples; a.flags = 0; a.vals_out = (float4 *)records_out;
    a.pts_c = (const float4 *)pts_c; a.dirs_c = (const float4 *)dirs_c;
    a.act = act; a.act_stride = act_stride; a.act_off = act_off;
    const unsigned groups = (unsigned)((n_rays + 255) / 256);
    const unsigned splits = sample_splits(groups, n_samples);
    a.s_per = (n_samples + (int)splits - 1) / (int)splits;
    hipLaunchKernelGGL((k_march<true, true, 8, true, true>), dim3(groups, (unsigned)((n_samples + a.s_per - 1) / a.s_per)), dim3(512), 0,
                       (hipStream_t)stream, a);
    return hl::check_launch("k_march<eval points, acts>");
}

// (per-workgroup maxima of k_df_transpose: 27 per tile of 32 rays)
size_t hl_render_plane_grads_scratch_bytes(int64_t n_rays) { return ((size_t)tiles32(n_rays) * 27 * sizeof(unsigned) + 255) / 256 * 256; }

size_t hl_render_plane_grads_points_scratch_bytes(int64_t n_rays, int n_samples, int n_importance) {
    return hl_render_plane_grads_scratch_bytes(n_rays) + (size_t)n_rays * (size_t)((n_samples + 63) / 64 + (n_importance + 63) / 64) * 6 * sizeof(float) + 256;
}

int hl_render_plane_grads_points(int H, int W, const float *bounds, const float *pts_coarse, const float *pts_new, int64_t n_rays,
                                 int n_samples, int n_importance, const float *del, int64_t del_stride, float *d_planes, void *scratch,
                                 void *stream) {
    HL_REQUIRE(bounds && pts_coarse && pts_new && del && d_planes && scratch, "hl_render_plane_grads_points: null argument");
    HL_REQUIRE(H > 0 && W > 0 && n_rays > 0 && n_samples >= 1 && n_importance >= 1, "hl_render_plane_grads_points: bad sizes");
    HL_REQUIRE(del_stride >= tiles32(n_rays) * 32 * (int64_t)(n_samples + n_importance), "hl_render_plane_grads_points: delta rows too short");
    DfTransposeArgs tr{const_cast<float *>(del), del_stride, n_rays, n_samples, n_importance, (unsigned *)scratch};
    hipLaunchKernelGGL(k_df_transpose, dim3(27, (unsigned)tiles32(n_rays)), dim3(256), 0, (hipStream_t)stream, tr);
    ScatterPtsArgs a{{(const float4 *)pts_coarse, (const float4 *)pts_new}, bounds, n_rays, n_samples, n_importance, H, W, del, del_stride,
                     d_planes, reinterpret_cast<float *>(static_cast<char *>(scratch) + hl_render_plane_grads_scratch_bytes(n_rays)), (const unsigned *)scratch};
    const long long nblk = n_rays * ((n_samples + 63) / 64 + (n_importance + 63) / 64);
    hipLaunchKernelGGL(k_block_bbox, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    int rcode = hl::check_launch("k_block_bbox");
    if (rcode) return rcode;
    const unsigned tiles = (unsigned)(((W + SC_TILE - 1) / SC_TILE) * ((H + SC_TILE - 1) / SC_TILE));
    hipLaunchKernelGGL(k_plane_scatter_pts, dim3(9, tiles), dim3(SC_THREADS), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_plane_scatter_pts");
}

static int64_t wgrad_points_per_range(int64_t n_cols) {   // 256 point ranges (HL_WGRAD_RANGES: developer knob, read once) of at least 1024 points
    static const int64_t nr = [] { const char *e_ = getenv("HL_WGRAD_RANGES"); return e_ ? (int64_t)atol(e_) : (int64_t)256; }();
    int64_t per = ((n_cols + nr - 1) / nr + 31) / 32 * 32;
    return per < 1024 ? 1024 : per;
}
static void wgrad_jobs(WgradArgs &a, const hl_render_mlp_grads *g) {
    a.job[0] = WgradJob{DROW_X0, 128, ROW_F, 27, 0, 27, 1, g ? g->pts0_w : nullptr, g ? g->pts0_b : nullptr};
    a.job[1] = WgradJob{DROW_X1, 128, ROW_X0, 128, 0, 128, 1, g ? g->pts1_w : nullptr, g ? g->pts1_b : nullptr};
    a.job[2] = WgradJob{DROW_X2, 128, ROW_F, 155, 0, 155, 1, g ? g->pts2_w : nullptr, g ? g->pts2_b : nullptr};       // input = [features | hidden1] = rows 0..154
    a.job[3] = WgradJob{DROW_Y, 128, ROW_X2, 128, 0, 128, 1, g ? g->feat_w : nullptr, g ? g->feat_b : nullptr};
    a.job[4] = WgradJob{DROW_V, 64, ROW_Y, 155, 0, 155, 1, g ? g->views_w : nullptr, g ? g->views_b : nullptr};        // input = [feature | view encoding]
    // heads, operands swapped: A = activation rows (128 / 64), B = their 1 / 3 delta rows; C[i][n] -> weight[n][i]
    a.job[5] = WgradJob{ROW_X2, 128, DROW_REC, 1, 1, 1, 128, g ? g->alpha_w : nullptr, g ? g->alpha_b : nullptr};
    a.job[6] = WgradJob{ROW_V, 64, DROW_REC + 1, 3, 1, 1, 64, g ? g->rgb_w : nullptr, g ? g->rgb_b : nullptr};
    int off = 0;
    for (int j = 0; j < WGRAD_JOBS; ++j) {
        a.part_off[j] = off;
        off += a.job[j].M * a.job[j].N + (a.job[j].a_is_act ? a.job[j].N : a.job[j].M);      // the products, then the bias sums (rows of the delta operand)
    }
    a.part_len = off;
}

size_t hl_render_weight_grads_scratch_bytes(int64_t n_cols) {
    if (n_cols <= 0) return 0;
    WgradArgs a{};
    wgrad_jobs(a, nullptr);
    const int64_t per = wgrad_points_per_range(n_cols);
    return (size_t)((n_cols + per - 1) / per) * a.part_len * sizeof(float);
}

int hl_render_weight_grads(const float *del, int64_t del_stride, const float *act, int64_t act_stride, int64_t n_cols,
                           const hl_render_mlp_grads *g, void *scratch, void *stream) {
    HL_REQUIRE(del && act && g && scratch && n_cols > 0 && n_cols % 32 == 0 && del_stride >= n_cols && act_stride >= n_cols,
               "hl_render_weight_grads: bad argument");
    HL_REQUIRE((int64_t)ACT_ROWS * act_stride * 4 < (1LL << 31) && (int64_t)DEL_ROWS * del_stride * 4 < (1LL << 31),
               "hl_render_weight_grads: activation / delta matrices must stay below 2 GiB");
    WgradArgs a{};
    wgrad_jobs(a, g);
    for (int j = 0; j < WGRAD_JOBS; ++j) HL_REQUIRE(a.job[j].out && a.job[j].bias_out, "hl_render_weight_grads: null gradient pointer %d", j);
    a.del = del; a.act = act; a.del_stride = del_stride; a.act_stride = act_stride; a.n_cols = n_cols;
    const int64_t per = wgrad_points_per_range(n_cols);
    a.k_per_wg = (int)per;
    a.part = static_cast<float *>(scratch);
    const int n_ranges = (int)((n_cols + per - 1) / per);
    hipLaunchKernelGGL(k_wgrad, dim3(WGRAD_JOBS, (unsigned)n_ranges), dim3(256), 0, (hipStream_t)stream, a);
    int rc = hl::check_launch("k_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(k_wgrad_finish, dim3((unsigned)((a.part_len + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, n_ranges);
    return hl::check_launch("k_wgrad_finish");
}

int hl_render_rays(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                   const float *rays_o, const float *rays_d, const float *near, const float *far, const float *z_vals,
                   const float *u, int64_t n_rays, int n_samples, int n_importance, unsigned flags, float *rgb,
                   float *acc, float *depth, void *workspace, void *stream) {
    return hl_render_rays_u_event(mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far, z_vals, u, nullptr, n_rays, n_samples,
                                  n_importance, flags, rgb, acc, depth, workspace, stream);
}

int hl_render_rays_u_event(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                           const float *rays_o, const float *rays_d, const float *near, const float *far, const float *z_vals,
                           const float *u, void *u_ready_event, int64_t n_rays, int n_samples, int n_importance, unsigned flags, float *rgb,
                           float *acc, float *depth, void *workspace, void *stream) {
    // u_ready_event (hipEvent_t or null): `u` is being written on ANOTHER stream (hl_mt19937_uniform continuing the CPU generator); the
    // stream waits for it only in front of the importance-sampling launch - the coarse pass does not read u and runs next to the generator
    auto wait_u = [&]() -> int {
        if (u_ready_event) HL_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)u_ready_event, 0));
        return HL_OK;
    };
    if (n_importance > 0) {
        // renderer.py:250 reshapes the coarse densities to (.., n_importance): only equal counts are valid
        HL_REQUIRE(n_importance == n_samples, "render: n_importance (%d) must equal n_samples (%d)", n_importance,
                   n_samples);
        HL_REQUIRE(u && workspace, "render: u and workspace are required when n_importance > 0");
        if (!(flags & HL_RENDER_REEVALUATE)) {
            // Evaluate-once pipeline.  The reference runs the network on all n_samples + n_importance sorted depths in the fine pass
            // (renderer.py:252-256), i.e. re-evaluates the coarse points it already has densities for; the MLP output at a point
            // depends only on (ray, depth), so here every point is evaluated exactly once with the full MLP - coarse points in pass
            // A, the new ones in pass B - and k_composite merges the two sorted lists: same values, same compositing order,
            // bit-identical images, 23 % fewer FLOPs (256 x 132 608 instead of 128 x 79 616 + 256 x 132 608 per ray).
            const size_t T32 = (size_t)tiles32(n_rays) * 32;
            float *vc = (float *)workspace, *vn = vc + T32 * n_samples * 4;
            float *zn = vn + T32 * n_importance * 4;
            const int h16 = (flags & HL_RENDER_MLP_FP16) ? 1 : ((flags & HL_RENDER_MLP_FP16X2) ? 3 : ((flags & HL_RENDER_MLP_BF16X3) ? 2 : 0));
            int rcode = render_eval_impl(mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far, z_vals, 0, n_rays, n_samples,
                                         vc, h16, stream);
            if (rcode) return rcode;
            rcode = wait_u();
            if (rcode) return rcode;
            // round 6: two launches per view - the fine launch draws its own importance depths and composites as it evaluates (no fine records, no merge kernel)
            // (fp16x2 products only: with the bf16x3 planes the fused kernel spills - measured 41.1 against 40.1 ms per view)
            if (!(flags & HL_RENDER_FOUR_LAUNCH) && h16 == 3 && n_samples <= 128 && z_vals == nullptr) {
                return render_onepass_fine(mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far, vc, u, zn, n_rays, n_samples, flags, 2,
                                           rgb, acc, depth, stream);
            }
            rcode = hl_render_importance_new(vc, rays_d, near, far, z_vals, u, n_rays, n_samples, n_importance, zn, stream);
            if (rcode) return rcode;
            rcode = render_eval_impl(mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far, zn, 1, n_rays, n_importance, vn,
                                     h16, stream);
            if (rcode) return rcode;
            return hl_render_composite(near, far, z_vals, zn, vc, vn, n_rays, n_samples, n_importance, flags, rgb, acc, depth, stream);
        }
        float *sigma = (float *)workspace;
        float *z_all = sigma + (size_t)tiles32(n_rays) * 32 * n_samples;
        int rcode = hl_render_coarse(mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far, z_vals, n_rays,
                                     n_samples, sigma, stream);
        if (rcode) return rcode;
        rcode = wait_u();
        if (rcode) return rcode;
        rcode = hl_render_importance(sigma, rays_d, near, far, z_vals, u, n_rays, n_samples, n_importance, z_all, stream);
        if (rcode) return rcode;
        return render_fine_impl(mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far, z_all, 1, n_rays,
                                n_samples + n_importance, flags, rgb, acc, depth, stream);
    }
    return render_fine_impl(mlp_packed, planes_packed, H, W, bounds, rays_o, rays_d, near, far, z_vals, 0, n_rays, n_samples,
                            flags, rgb, acc, depth, stream);
}

int hl_camera_rays(const double *h_Kinv, const double *h_R, const double *h_T, const double *h_bounds, int H, int W,
                   float *rays_o, float *rays_d, float *near, float *far, unsigned char *mask_at_box, void *stream) {
    HL_REQUIRE(h_Kinv && h_R && h_T && h_bounds && rays_o && rays_d && near && far, "hl_camera_rays: null argument");
    HL_REQUIRE(H > 0 && W > 0, "hl_camera_rays: bad image size %dx%d", H, W);
    CamArgs a{};
    for (int i = 0; i < 9; ++i) { a.Ki[i] = h_Kinv[i]; a.R[i] = h_R[i]; }
    for (int c = 0; c < 3; ++c) {
        a.T[c] = h_T[c];
        a.o[c] = -((h_R[0 * 3 + c] * h_T[0] + h_R[1 * 3 + c] * h_T[1]) + h_R[2 * 3 + c] * h_T[2]);
        a.b[c] = h_bounds[c] + -0.01;
        a.b[3 + c] = h_bounds[3 + c] + 0.01;
    }
    a.H = H; a.W = W; a.rays_o = rays_o; a.rays_d = rays_d; a.near = near; a.far = far; a.mask = mask_at_box;
    const long long n = (long long)H * W;
    hipLaunchKernelGGL(k_camera_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_camera_rays");
}

int hl_deform_points(const float *pts, const float *dirs, const float *h_R, const float *h_Th, const float *verts_smpl4,
                     const float *table, int n_vertices, int64_t n_points, float *can_pts, float *can_dirs, int *vertex_ids,
                     void *stream) {
    HL_REQUIRE(pts && h_R && h_Th && verts_smpl4 && table && can_pts, "hl_deform_points: null argument");
    HL_REQUIRE(!dirs || can_dirs, "hl_deform_points: can_dirs is required when dirs are given");
    HL_REQUIRE(n_vertices > 0 && n_points > 0, "hl_deform_points: bad sizes");
    DeformArgs a{};
    a.pts = pts; a.dirs = dirs;
    for (int i = 0; i < 9; ++i) a.c.R[i] = h_R[i];
    for (int i = 0; i < 3; ++i) a.c.Th[i] = h_Th[i];
    a.c.verts = (const float4 *)verts_smpl4; a.c.table = table; a.c.V = n_vertices; a.P = n_points;
    a.can_pts = can_pts; a.can_dirs = can_dirs; a.vid = vertex_ids;
    hipLaunchKernelGGL(k_deform_points, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_deform_points");
}

int hl_deform_rays(const float *rays_o, const float *rays_d, const float *near, const float *far, const float *z, int z_tiled,
                   int64_t n_rays, int n_samples, const float *h_R, const float *h_Th, const float *verts_smpl4, const float *table,
                   int n_vertices, float *pts_c, float *dirs_c, void *scratch, void *stream) {
    HL_REQUIRE(rays_o && rays_d && near && far && h_R && h_Th && verts_smpl4 && table && pts_c && dirs_c, "hl_deform_rays: null argument");
    HL_REQUIRE(n_rays > 0 && n_samples >= 1 && n_vertices > 0, "hl_deform_rays: bad sizes");
    DeformRaysArgs a{};
    for (int i = 0; i < 9; ++i) a.c.R[i] = h_R[i];
    for (int i = 0; i < 3; ++i) a.c.Th[i] = h_Th[i];
    a.c.verts = (const float4 *)verts_smpl4; a.c.table = table; a.c.V = n_vertices;
    a.rays_o = rays_o; a.rays_d = rays_d; a.near = near; a.far = far; a.z = z; a.z_tiled = z_tiled; a.R = n_rays; a.S = n_samples;
    a.pts_c = (float4 *)pts_c; a.dirs_c = (float4 *)dirs_c;
    static const bool brute = getenv("HL_DEFORM_BRUTE") != nullptr;    // developer switch: full scan for every sample point
    if (n_vertices <= DC_MAXV && !brute) {
        const size_t lds = (size_t)(DC_MAXV + DC_MAXV / 64 + DC_WAVES * DC_SLOTS * 2) * sizeof(float4);
        static bool attr_set = false;
        if (!attr_set) {
            HL_HIP(hipFuncSetAttribute((const void *)k_deform_rays_cull, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        const long long items = (long long)tiles32(n_rays) * 2 * ((n_samples + 3) / 4);
        const unsigned grid = (unsigned)(items / DC_WAVES + 1 < 256 ? items / DC_WAVES + 1 : 256);
        HL_REQUIRE(scratch, "hl_deform_rays: scratch (8 bytes of device memory) is required");
        a.counter = (unsigned long long *)scratch;
        HL_HIP(hipMemsetAsync(a.counter, 0, sizeof(unsigned long long), (hipStream_t)stream));
        hipLaunchKernelGGL(k_deform_rays_cull, dim3(grid), dim3(DC_WAVES * 64), lds, (hipStream_t)stream, a);
        return hl::check_launch("k_deform_rays_cull");
    }
    const long long total = (long long)tiles32(n_rays) * 32 * ((n_samples + 1) / 2);   // two samples per thread
    hipLaunchKernelGGL(k_deform_rays, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_deform_rays");
}

int hl_render_eval_points(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *pts_c,
                          const float *dirs_c, int64_t n_rays, int n_samples, float *records_out, void *stream) {
    HL_REQUIRE(mlp_packed && planes_packed && bounds && pts_c && dirs_c && records_out, "hl_render_eval_points: null argument");
    HL_REQUIRE(n_rays > 0 && n_samples >= 1 && H > 0 && W > 0, "hl_render_eval_points: bad sizes");
    MarchArgs a{};
    a.packed = (const float *)mlp_packed; a.planes = (const float4 *)planes_packed; a.H = H; a.W = W; a.bounds = bounds;
    // rays are not read in POINTS mode beyond the placeholders below (the kernel loads o, d, near, far of its ray unconditionally)
    a.rays_o = pts_c; a.rays_d = pts_c; a.near = pts_c; a.far = pts_c;
    a.z = nullptr; a.z_tiled = 0; a.R = n_rays; a.S = n_samples; a.flags = 0; a.vals_out = (float4 *)records_out;
    a.pts_c = (const float4 *)pts_c; a.dirs_c = (const float4 *)dirs_c;
    hipLaunchKernelGGL((k_march<true, true, 8, true>), dim3((unsigned)((n_rays + 255) / 256)), dim3(512), 0, (hipStream_t)stream, a);
    return hl::check_launch("k_march<eval points>");
}

size_t hl_render_canonical_workspace_bytes(int64_t n_rays, int n_samples, int n_importance) {
    if (n_rays <= 0) return 256;
    const size_t T32 = (size_t)tiles32(n_rays) * 32;
    const int smax = n_samples > n_importance ? n_samples : n_importance;
    return hl_render_workspace_bytes(n_rays, n_samples, n_importance > 0 ? n_importance : n_samples) + T32 * smax * 2 * sizeof(float4) + 512;
}

int hl_render_rays_canonical(const void *mlp_packed, const void *planes_packed, int H, int W, const float *t_bounds,
                             const float *rays_o, const float *rays_d, const float *near, const float *far, const float *z_vals,
                             const float *u, int64_t n_rays, int n_samples, int n_importance, unsigned flags, const float *h_R,
                             const float *h_Th, const float *verts_smpl4, const float *table, int n_vertices, float *rgb, float *acc,
                             float *depth, void *workspace, void *stream) {
    HL_REQUIRE(workspace && n_rays > 0 && n_samples >= 2, "hl_render_rays_canonical: bad argument");
    HL_REQUIRE(n_importance == 0 || n_importance == n_samples, "render: n_importance (%d) must equal n_samples (%d)", n_importance,
               n_samples);
    HL_REQUIRE(n_importance == 0 || u, "render: u is required when n_importance > 0");
    HL_REQUIRE(n_importance > 0, "hl_render_rays_canonical: n_importance = 0 is not built (the reference's configurations use 128)");
    const size_t T32 = (size_t)tiles32(n_rays) * 32;
    float *vc = (float *)workspace, *vn = vc + T32 * n_samples * 4;
    float *zn = vn + T32 * n_importance * 4;
    float *pc = zn + T32 * n_importance;                       // canonical points of the pass in flight, then directions
    pc = (float *)(((uintptr_t)pc + 255) / 256 * 256);
    const int smax = n_samples > n_importance ? n_samples : n_importance;
    float *dc = pc + T32 * smax * 4;
    void *dscr = dc + T32 * smax * 4;                           // work counter of the deformation kernel
    int rc = hl_deform_rays(rays_o, rays_d, near, far, z_vals, 0, n_rays, n_samples, h_R, h_Th, verts_smpl4, table, n_vertices, pc, dc,
                            dscr, stream);
    if (rc) return rc;
    rc = hl_render_eval_points(mlp_packed, planes_packed, H, W, t_bounds, pc, dc, n_rays, n_samples, vc, stream);
    if (rc) return rc;
    rc = hl_render_importance_new(vc, rays_d, near, far, z_vals, u, n_rays, n_samples, n_importance, zn, stream);
    if (rc) return rc;
    rc = hl_deform_rays(rays_o, rays_d, near, far, zn, 1, n_rays, n_importance, h_R, h_Th, verts_smpl4, table, n_vertices, pc, dc,
                        dscr, stream);
    if (rc) return rc;
    rc = hl_render_eval_points(mlp_packed, planes_packed, H, W, t_bounds, pc, dc, n_rays, n_importance, vn, stream);
    if (rc) return rc;
    return hl_render_composite(near, far, z_vals, zn, vc, vn, n_rays, n_samples, n_importance, flags, rgb, acc, depth, stream);
}

}  // extern "C"
