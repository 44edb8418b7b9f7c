// Launchers of the UNet / sampler device kernels (hl_unet_kernels.hip). Internal header.
#pragma once
#include "hl_common.h"

namespace hl {

// NHWC fp32 tensor view with a channel pitch (so two producers can write into one "concat" buffer).
struct View {
    float *p = nullptr;
    int N = 0, H = 0, W = 0, C = 0;
    long pitch = 0;  // floats between consecutive pixels
    long pixels() const { return (long)N * H * W; }
};

// Where a kernel that normalises its input gets the per-(image, channel) affine y = x*A + B of GroupNorm32 [+ scale/shift] from: the
// arrays cA / cB (groupnorm_coef / groupnorm_coef_stats wrote them), or - gt set - the fixed-point group totals its producer(s) left
// (ConvArgs::stats) plus the norm's own parameters: every consumer workgroup then forms the coefficients of its image itself
// (coef_to_lds, hl_stats.h) and no coefficient launch sits between producer and consumer.  C = channels of the normalised view, HW =
// pixels per image (it also fixes the number of shards of the totals).
struct GnSrc {
    const float *gt;
    int C, HW;
    const float *gamma, *beta, *emb;   // emb (N, emb_pitch) holds [scale(C) | shift(C)] or null
    long emb_pitch;
    float eps;
};

struct ConvArgs {
    View in;             // C must be a multiple of 16 (pad channels are zero and have zero weights)
    const float *w;      // packed [Cout_pad][Ktot], K order = kt_decode() in hl_unet_kernels.hip (groups of two 16-channel chunks, taps inside)
    const void *w_bf3;   // optional: the same weights split into three bf16 planes (conv_pack_weights_bf3); selects k_conv_bf3
    int bf16_single;     // with w_bf3: 1 = HL_CONV_BF16 (activations rounded to bf16 x the weights' two leading bf16 planes), 0 = bf16x3 emulation
    const void *w_h2;    // optional (1x1 layers): two fp16 planes in MFMA-fragment order (conv_pack_weights_h2); selects k_conv1_h2 where it fills the chip
    const float *in_absmax; // optional, instead of in_stats: [N] the largest |x| of every image of `in` (tensor_absmax)
    const float *in_stats; // optional: the group totals the producer(s) of `in` left for ANY view that covers it (conv_stats_floats(N, in.H * in.W) floats, complete
                         // when this launch starts): the fp16x2 kernels derive the power-of-two scale of the raw input from sum x^2 (ConvK::xs_gt); null: scale 1
    const void *w_h16;   // optional: 16-bit weights in MFMA-fragment order (conv_pack_weights_h16); selects k_conv_h16 where conv_h16_applies
    int h16_fp16;        // with w_h16: 1 = fp16 operands (HL_CONV_FP16), 0 = bf16 (HL_CONV_BF16)
    const float *w_wino; // optional: Winograd-domain weights (conv_pack_weights_wino); selects k_conv_wino for large 3x3 layers
    const float *w_wino4;// optional: Winograd F(4x4,3x3) weights (conv_pack_weights_wino4); selects k_conv_wino4 where it fills the chip
    const float *bias;   // [Cout] or null
    int Cout;            // real output channels
    int ks;              // 1 or 3 (pad = ks/2)
    int stride;          // 1 or 2
    int ups;             // 1: input is read through a nearest x2 upsample (unet.py:77)
    const float *coefA;  // per-(n, cin) affine applied before the conv (GroupNorm [+scale/shift]) or null
    const float *coefB;
    GnSrc gn;            // alternative to coefA / coefB (gn.t0 != null): the affine formed in the kernels from the producers' totals
    int act;             // 1: SiLU after the affine
    View out;            // NHWC (pitch honoured) unless out_nchw
    const float *res;    // residual added to out (may alias out.p), pitch res_pitch; or null
    long res_pitch;
    float *out2;         // optional second output out2 = out + res2
    long out2_pitch;
    const float *res2;
    long res2_pitch;
    int out_nchw;        // write (N, Cout, H, W) instead of NHWC
    float *act_ws;       // optional scratch (pixels*Cin floats) for the materialised GroupNorm(+SiLU) input of k_conv_dma
    size_t act_ws_bytes;
    mutable int path;    // set by conv2d: 0 direct implicit GEMM, 1 Winograd F(2x2,3x3), 2 bf16x3 emulation, 3 Winograd F(4x4,3x3), 5 k_conv_h16, 6 k_conv1_h2
    float *splitk_ws;    // optional scratch for split-K partial sums (small-M layers); null disables split-K
    size_t splitk_ws_bytes;
    // GroupNorm statistics of the OUTPUT for the layer that will normalise it, emitted by the epilogue of whichever kernel stores
    // the tensor: per image and GROUP of the normalised view the totals (sum x, sum x^2), as 64-bit FIXED-POINT integers [shard][N][32][2]
    // that the workgroups add to with integer atomics (hl_stats.h) - integer addition is associative, so the totals are bit-identical
    // from run to run.  `stats` (and `stats2` for out2) need conv_stats_floats(N, pixels per image) floats each, ZEROED before the first
    // producer of the view runs; null = not wanted.  st_cg = channels per group of the view (0: Cout / 32), st_c0 = channel of the view
    // that output channel 0 is (a decoder "concat" has two producers adding into one block).  On return stat_slots is 1 when the launch
    // added its totals, 0 when this path emits none (register-staged / bf16x3 / NCHW / odd sizes): the consumer then computes the
    // statistics from the tensor (groupnorm_coef).
    float *stats, *stats2;
    int st_cg, st_c0, st2_cg, st2_c0;
    mutable int stat_slots;
    hipEvent_t ev_mid;   // optional (profiling): recorded between the GroupNorm pre-pass and the convolution kernel, when there is a pre-pass
    mutable int ev_mid_used;
    int plan_only;       // 1: no launch, only set `path` (w / w_wino / w_bf3 are then just non-null markers of what could be packed)
};
// kernel-side argument block of the convolution kernels (filled by conv2d)
struct ConvK {
    const float *in; long in_pitch; int N, Hin, Win, Cin;
    int Hout, Wout, ks, stride, ups, taps;
    const float *w; const void *w_bf3; const float *w_wino; long Ktot; const float *bias; int Cout;
    const float *cA; const float *cB; int act;
    GnSrc gn;          // (cA null and gn.t0 set: the coefficients are formed in the kernel, coef_to_lds)
    float *out; long out_pitch; const float *res; long res_pitch;
    float *out2; long out2_pitch; const float *res2; long res2_pitch;
    int out_nchw; long M; int wrows;
    int kt_per;        // k-tiles per split (blockIdx.z); gridDim.z == 1 -> whole K
    int n_mtiles, n_nblocks;
    float *partial;    // split-K: raw accumulators [z][M][Cout]
    // GroupNorm statistics of the OUTPUT, emitted by the epilogue for the layer that will normalise it (nn.py:17-19): fixed-point totals
    // (sum, sum of squares) of the stored value (after bias / residual) per image and group of the normalised view, [shard][N][32][2]
    // 64-bit integers the epilogues add to atomically (hl_stats.h); null = not wanted.  st2: the same for out2.  cg / c0: channels per
    // group of the view, the view channel of output channel 0.  Deterministic (integer adds commute).
    int st1_cg, st1_c0, st2_cg, st2_c0;
    float *st1, *st2;
    int in16;          // k_conv_h16 / k_conv1_h16: `in` holds 16-bit values (the GroupNorm pass wrote them), in_pitch counts them
    // fp16x2 kernels (round 6: the split products are scale-invariant).  wsc: [Cout] inverse power-of-two scales of the weight planes (conv_h2_wscale) - the
    // epilogue multiplies the accumulators by them.  xs_gt: group totals of the raw INPUT tensor (the block its producer(s) left, ConvArgs::in_stats) or null:
    // the staging multiplies the input by the power of two sx with |x| sx <= sqrt(sum x^2) sx <= 32752 (no overflow of the fp16 planes, and the low plane stays
    // normal for values down to 2^-11 of the image's largest possible), the epilogue by 1 / sx; a fused GroupNorm takes its sx from the coefficient bound instead.
    const float *wsc;
    const float *xs_gt;
    int xs_hw;         // pixels per image of the tensor the totals belong to (fixes the number of shards)
    const float *xs_max;   // alternative to xs_gt: [N] the largest |x| of every image (tensor_absmax) - exact at any magnitude (the single-op entry points: gradients are 1e-4 ... 1e-9)
};

// k_conv_wino4w (hl_conv_wino4w.hip): Winograd F(4x4,3x3) with 64 output channels per workgroup, one wave per SIMD, 18 accumulator
// tiles per wave in the accumulator registers; reads the weights conv_pack_weights_wino4 laid out.  LDS: conv_wino4w_lds_bytes().
size_t conv_wino4w_lds_bytes();
int conv_wino4w_launch(const ConvK &p, int ups, int blk, int splits, hipStream_t st);

// k_conv_h16 (hl_conv_h16.hip): 3x3 / stride-1 convolution with 16-bit operands (fp16 / bf16) and fp32 accumulation, 16x16 pixels x 192
// channels per workgroup; reads ConvK::w_bf3 = the weights conv_pack_weights_h16 laid out (ConvK::n_mtiles = N (H/16)(W/16), n_nblocks = Cout/192).
bool conv_h16_applies(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int ups);
size_t conv_packed_h16_bytes(int Cout, int Cin_pad, int ks);
int conv_pack_weights_h16(const float *w_oihw, int Cout, int Cin, int Cin_pad, int ks, void *packed, int f16, hipStream_t st, int tf = 0);
int conv_h16_launch(const ConvK &p, int f16, hipStream_t st, int splits = 1);
// k_conv1_h2 (hl_conv_h16.hip): the 1x1 / stride-1 convolutions of the default fp32 mode with fp16x2 products (two fp16 planes per operand, three partial
// products, fp32 accumulation); weights conv_pack_weights_h2 laid out (ConvK::w_bf3), ConvK::n_mtiles = pixels / 256, n_nblocks = Cout / 192.
bool conv1_h2_applies(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int ups);
size_t conv_packed_h2_bytes(int Cout, int Cin_pad, int ks);
int conv_pack_weights_h2(const float *w_oihw, int Cout, int Cin, int Cin_pad, int ks, void *packed, hipStream_t st, int tf = 0);
const float *conv_h2_wscale(const void *packed, int Cout, int Cin_pad, int ks);   // the [Cout] inverse scales behind the planes (ConvK::wsc)
int conv1_h2_launch(const ConvK &p, hipStream_t st);
int conv1_h2s_launch(const ConvK &p, hipStream_t st, int splits = 1);               // the same on 128-pixel tiles, two workgroups per CU (p.n_mtiles = pixels / 128)
bool conv3_h2d_applies(int Hout, int Wout, int Cin, int Cout);        // 3x3 / stride 2 with fp16x2 products (k_conv_h2d)
int conv3_h2d_launch(const ConvK &p, hipStream_t st);                 // p.n_mtiles = output pixels / 128
int conv3_h2s_launch(const ConvK &p, hipStream_t st, int splits = 1);  // the same on 8x16-pixel tiles, two workgroups per CU (p.n_mtiles = pixels / 128)
int conv3_h2_launch(const ConvK &p, hipStream_t st, int splits = 1);   // 3x3 / stride 1: k_conv_h16's workgroups with two planes (ConvK::in16 = 2: a two-plane image from the GroupNorm pass)
void set_h16_min_blocks(long v);   // developer / test switch: workgroups from which the dispatch takes k_conv_h16 (< 0: the default, 48)

// Statistics block of one normalised VIEW (ConvArgs::stats): [shard][N][32 groups][2] 64-bit fixed-point totals.  Levels with many
// workgroups per image keep stat_shards(HW) = 8 copies (a workgroup adds to copy blockIdx.x % 8: atomics on one word serialise), the low
// levels one.
constexpr int stat_shards(long HW) { return HW >= 4096 ? 8 : 1; }
constexpr size_t conv_stats_floats(int N, long HW) { return (size_t)stat_shards(HW) * N * 32 * 4; }
size_t conv_splitk_ws_bytes();
// totals (conv_stats_floats(x.N, x.H * x.W) floats, zeroed here) of a tensor nobody left totals for: what ConvArgs::in_stats wants (the single-op entry points)
int tensor_totals(const View &x, float *totals, hipStream_t st);
// [N] floats: the largest |x| of every image (zeroed here; non-negative floats order like their bit patterns: one integer atomicMax per workgroup)
int tensor_absmax(const View &x, float *amax, hipStream_t st);
int conv2d(const ConvArgs &a, hipStream_t st);
size_t conv_packed_floats(int Cout, int Cin_pad, int ks);
// tf = 1: the source is laid out (Cin, Cout, ks, ks) and is read flipped and channel-transposed (backward-data weights)
int conv_pack_weights(const float *w_oihw, int Cout, int Cin, int Cin_pad, int ks, float *packed, hipStream_t st, int tf = 0);
// split-bf16 copy for k_conv_bf3: [Cout_pad][K/16][plane*2 + k-half][8 bf16], 6 bytes per weight; 0 bytes if the layer
// never takes the DMA tile (Cout_pad not a multiple of 96)
// Winograd F(2x2,3x3) copy U = G g G^T: [Cout/64][Cin_pad/8][16][2][2][32][4] floats; 0 bytes if not applicable
size_t conv_packed_wino_bytes(int Cout, int Cin_pad, int ks);
int conv_pack_weights_wino(const float *w_oihw, int Cout, int Cin, int Cin_pad, float *packed, hipStream_t st, int tf = 0);
// Winograd F(4x4,3x3) copy (points 0, +-3/4, +-3/2, inf): [Cout/32][Cin_pad/8][36][2][32][4] floats; 0 bytes if not applicable
size_t conv_packed_wino4_bytes(int Cout, int Cin_pad, int ks);
int conv_pack_weights_wino4(const float *w_oihw, int Cout, int Cin, int Cin_pad, float *packed, hipStream_t st, int tf = 0);
size_t conv_packed_bf3_bytes(int Cout, int Cin_pad, int ks);
int conv_pack_weights_bf3(const float *w_oihw, int Cout, int Cin, int Cin_pad, int ks, void *packed, hipStream_t st, int tf = 0);

// GroupNorm(32 groups, eps 1e-5) statistics -> per-(n,c) affine  y = x*A + B   (nn.py:17-19,100)
// optional scale/shift (ResBlock use_scale_shift_norm, unet.py:203-206): y = GN(x)*(1+scale)+shift,
// where emb (N, emb_pitch) holds [scale(C) | shift(C)] starting at emb + n*emb_pitch.
size_t gn_scratch_floats(int N);
// the same affine (as arrays) from the group totals the producer(s) of the view left (ConvArgs::stats)
int groupnorm_coef_stats(const View &x, const float *group_totals, const float *gamma, const float *beta, const float *emb,
                         long emb_pitch, float *coefA, float *coefB, hipStream_t st, float eps = 1e-5f);
// gstat (optional): (N, 32, 2) = (mean, rstd) of every group, for the backward pass of the training path
int groupnorm_coef(const View &x, const float *gamma, const float *beta, const float *emb, long emb_pitch, float *coefA,
                   float *coefB, float *scratch, hipStream_t st, float *gstat = nullptr, float eps = 1e-5f);

// y (dense NHWC) = act ? silu(x*A + B) : x*A + B with the per-(n,c) affine of groupnorm_coef (the pre-pass of the DMA convs; the
// GroupNorm forward of the training path)
int gn_apply(const View &x, const float *coefA, const float *coefB, int act, float *y, hipStream_t st);

// out[b][o] = bias[o] + sum_k act(in[b][k]) * W[o][k] (+ addrow[idx[b]][o]);  B <= 8
int linear_small(const float *in, long in_pitch, int B, int K, const float *W, const float *bias, int O, int silu_in,
                 const float *addrow, const int64_t *idx, float *out, long out_pitch, hipStream_t st);
// sinusoidal timestep embedding (nn.py:103-121); t is int64 (B,)
int timestep_embedding(const int64_t *t, const float *t_float, int B, int dim, float *out, hipStream_t st);

// QKV attention (unet.py:255-274): qkv (N, T, 3C) with channel = head*3ch + {q|k|v}*ch + c ; out (N, T, C)
// h2: fp16x2 products where a kernel has them (the default conv mode's attention).  out_totals: optional zeroed conv_stats_floats(N, T) floats - the key-split kernels add the
// sum x^2 of the output they store (the projection convolution's activation scale); *totals_emitted says whether the kernel that ran did
int attention(const float *qkv, int N, int T, int C, int heads, float *out, hipStream_t st, int h2 = 0, float *out_totals = nullptr, int *totals_emitted = nullptr);
size_t attention_backward_scratch_bytes(int N, int T, int C, int heads);                       // hl_attention_bwd.hip
int attention_backward(const float *qkv, const float *out, const float *dout, int N, int T, int C, int heads, float *dqkv, void *scratch,
                       size_t scratch_bytes, hipStream_t st);

// (B,C,H,W) x, x_cond -> NHWC padded to Cpad: x_nhwc and (x + x_cond)_nhwc   (unet.py:588,596)
// x_tot / xsum_tot: optional zeroed conv_stats_floats(B, H * W) floats each - the outputs' sum x^2 per image (the input convolutions' activation scale)
int prep_inputs(const float *x, const float *x_cond, int B, int C, int H, int W, int Cpad, float *x_nhwc, float *xsum_nhwc,
                hipStream_t st, float *x_tot = nullptr, float *xsum_tot = nullptr);

// cond_type='cross_attention' (spatial_transformer.py): nn.LayerNorm over the channels of every pixel -> y dense (pixels, C); GEGLU on
// (pixels, 2F) -> (pixels, F); x (N, HW, C) += v (N, C)
int layernorm(const View &x, const float *gamma, const float *beta, float *y, hipStream_t st);
int geglu(const float *in, long npix, int F, float *out, hipStream_t st);
int add_rowvec(const View &x, const float *v, hipStream_t st, long vpitch = 0);   // x[n, p, c] += v[n * vpitch + c] (vpitch 0: C)
// use_3d_aware=True (unet.py:566-570, 613-614): (B, 3C, H, W) NCHW <-> the planes side by side, NHWC (B, H, 3W, Cpad) / NCHW (B, C, H, 3W)
int prep_inputs_3d(const float *x, const float *x_cond, int B, int C, int H, int W, int Cpad, float *x_nhwc, float *xsum_nhwc,
                   hipStream_t st);
int unroll_planes(const float *rolled_nchw, int B, int C, int H, int W, float *out, hipStream_t st);
// ResBlock step of unet.py:208-214: y (N, H, 3W, 3C dense) = silu(cat[A h + B, two plane means of it]); sums: N*3*(H + W/3)*C floats
int gn_apply_3d(const View &h, const float *coefA, const float *coefB, float *sums, float *y, hipStream_t st);

}  // namespace hl
