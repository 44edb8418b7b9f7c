/*
 * humanliff_hip.h — C ABI of libhumanliff_hip.so (MI355X / gfx950 only).
 *
 * The reference (skhu101/HumanLiff) has no FFI layer: its boundary is the
 * Python API of two hot paths.  Each entry point below names the reference
 * interface it replaces (paths relative to /root/reference).  The Python
 * mirrors in humanliff_amd/ (same class / function names and argument
 * meaning as the reference) are thin callers of this ABI; INTEGRATION.md
 * shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named h_* (host);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - the library never allocates device memory: packed buffers and workspace
 *     are sized by *_bytes() and owned by the caller;
 *   - all calls are enqueue-only (no device synchronisation) and re-entrant
 *     per (device, stream);
 *   - return 0 on success, a negative hl_status on failure; the message is
 *     available (thread-local) from hl_last_error().
 */
#ifndef HUMANLIFF_HIP_H
#define HUMANLIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum hl_status {
    HL_OK = 0,
    HL_ERR_INVALID = -1,   /* bad argument (the reference would `assert`)        */
    HL_ERR_UNSUPPORTED = -2, /* configuration outside what the kernels cover      */
    HL_ERR_RUNTIME = -3    /* HIP runtime error (launch failed, wrong device ...) */
} hl_status;

int hl_version(void);
const char *hl_last_error(void);

/* ------------------------------------------------------------------------
 * Path 2 — tri-plane NeRF volume renderer
 * ------------------------------------------------------------------------ */

/* Render-MLP parameters, in the reference's state_dict layout (row-major
 * (out,in) nn.Linear weights), human_diffusion/NeRF/renderer.py:29-39. */
typedef struct hl_render_mlp_params {
    const float *pts0_w, *pts0_b;   /* pts_linears.0   (128,27)  (128) */
    const float *pts1_w, *pts1_b;   /* pts_linears.1   (128,128) (128) */
    const float *pts2_w, *pts2_b;   /* pts_linears.2   (128,155) (128) */
    const float *feat_w, *feat_b;   /* feature_linear  (128,128) (128) */
    const float *alpha_w, *alpha_b; /* alpha_linear    (1,128)   (1)   */
    const float *views_w, *views_b; /* views_linear    (64,155)  (64)  */
    const float *rgb_w, *rgb_b;     /* rgb_linear      (3,64)    (3)   */
} hl_render_mlp_params;

/* Re-lay the MLP for the MFMA ray-march kernel (done once per checkpoint). */
size_t hl_render_mlp_packed_bytes(void);
int hl_render_mlp_pack(const hl_render_mlp_params *h_params, void *packed, void *stream);

/* Re-lay one subject's tri-plane (3,9,H,W) fp32 — the (1,3,9,H,W) tensor the
 * sampler script builds at scripts/triplane_sample_layered.py:158 — into the
 * texel-major layout the kernel gathers from (done once per subject). */
size_t hl_planes_packed_bytes(int H, int W);
int hl_planes_pack(const float *planes, int H, int W, void *packed, void *stream);

#define HL_RENDER_WHITE_BKGD 1u      /* renderer.py:224-225 (per-ray intent)            */
#define HL_RENDER_NORMALIZE_DEPTH 2u /* depth = (depth - near) / (far - near + 1e-5): NeRF/renderer.py:271, recon_NeRF/lib/renderer.py:288 */
#define HL_RENDER_REEVALUATE 4u      /* hl_render_rays: run the fine pass over all n_samples+n_importance depths like the reference
                                        (re-evaluating the coarse points) instead of evaluating every point once; same image */
#define HL_RENDER_CLAMP_DEPTH 8u     /* clamp the normalised depth to [0,1]: NeRF/renderer.py:272-274 - the human_diffusion twin only,
                                        recon_NeRF/lib/renderer.py leaves it unclamped */

#define HL_RENDER_MLP_FP16 16u       /* opt-in (hl_render_rays, evaluate-once pipeline): the MLP with fp16 operands / fp32 accumulation
                                        (k_march16, all weights LDS-resident); features, encodings, softplus, compositing stay fp32 */
#define HL_RENDER_MLP_BF16X3 32u     /* hl_render_rays, evaluate-once pipeline: every fp32 product of the MLP (renderer.py:134-156) formed from an
                                        EXACT three-way bf16 split of both operands - six partial products on v_mfma_f32_32x32x16_bf16, fp32
                                        accumulation, dropped terms < 2^-24 |a b| (k_march_b3); an fp32-tolerance mode, not a reduced-precision one */

#define HL_RENDER_MLP_FP16X2 64u     /* the same pipeline with TWO fp16 planes per operand (activations x = h0 + h1 to 2^-20 |x| while 2^-3 <= |x| < 65504, absolute 2^-24
                                        below; weights: every layer's planes are those of 2^k W with max |2^k W| in [2^12, 2^13) - nearest-even, 2^-22 of the layer's
                                        largest weight WHATEVER its magnitude (round 6; the accumulators are multiplied by 2^-k on their way into the softplus) - and
                                        the three partial products h0 w0 + h0 w1 + h1 w0 on v_mfma_f32_32x32x16_f16, fp32 accumulation (k_march_plw<2>): half the
                                        matrix instructions of BF16X3 at the same fp32-class error against the reference's renders (goldens a ... f: weights from
                                        2^-8 below to 2^4 above nn.Linear's initialisation); softplus returns x beyond a pre-activation of 88.7 like F.softplus */

#define HL_RENDER_FOUR_LAUNCH 128u    /* hl_render_rays: keep rounds 2-5's schedule of the evaluate-once pipeline - evaluate, k_importance, evaluate, k_composite, with the raw
                                        records of BOTH halves through HBM.  Default (round 6, HL_RENDER_MLP_FP16X2, n_samples <= 128, linspace depths): TWO
                                        launches per view - the coarse evaluate, then the one-pass fine launch (k_march_plw<., false, true>): the wave that owns 32
                                        rays draws their importance depths from the coarse records (renderer.py:158-170, 533-563), evaluates them and composites
                                        coarse + new samples in depth order as it goes (renderer.py:172-231, 252-253); bit-identical images */

size_t hl_render_workspace_bytes(int64_t n_rays, int n_samples, int n_importance);

/* Replaces Renderer.render (human_diffusion/NeRF/renderer.py:234-281,
 * recon_NeRF/lib/renderer.py:244-291) together with the per-chunk body of
 * render() (scripts/triplane_sample_layered.py:262-279,
 * recon_NeRF/run_nerf_batch.py:41-58) for use_canonical_space=False, test mode.
 *
 *   bounds   (2,3)   tp_input['world_bounds'][b]
 *   rays_o/d (R,3)   near/far (R)
 *   z_vals   (R,n_samples) or NULL -> near*(1-t)+far*t, t = linspace(0,1,n_samples)
 *   u        (R,n_importance) uniform draws of sample_pdf (renderer.py:545); required
 *            when n_importance > 0 (n_importance must then equal n_samples, renderer.py:250)
 *   rgb (R,3)  acc (R)  depth (R)   outputs; normal_map == rgb_map in the reference
 *   workspace  hl_render_workspace_bytes(); scratch, tile-major [ceil(R/32)][sample][32 rays] (ray r = tile r/32, lane r%32: one
 *            line per wave and sample instead of 4-byte accesses at a row stride): the raw (sigma, r, g, b) records of the
 *            n_samples coarse and the n_importance new points and the sorted new depths - every point is evaluated ONCE with the
 *            full MLP and a merge kernel composites them in depth order (the reference re-evaluates the coarse points in its
 *            fine pass; the outputs are bit-identical, HL_RENDER_REEVALUATE selects that schedule)
 */
int hl_render_rays(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                   const float *rays_o, const float *rays_d, const float *near, const float *far,
                   const float *z_vals, const float *u, int64_t n_rays, int n_samples, int n_importance,
                   unsigned flags, float *rgb, float *acc, float *depth, void *workspace, void *stream);
/* hl_render_rays whose `u` is still being written on another stream (hl_mt19937_uniform): u_ready_event (hipEvent_t, recorded behind that
 * writer; NULL = none) is waited for in front of the importance-sampling launch only, so the coarse evaluate pass overlaps the generator. */
int hl_render_rays_u_event(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                   const float *rays_o, const float *rays_d, const float *near, const float *far,
                   const float *z_vals, const float *u, void *u_ready_event, int64_t n_rays, int n_samples, int n_importance,
                   unsigned flags, float *rgb, float *acc, float *depth, void *workspace, void *stream);

/* The three stages of hl_render_rays, exposed for tests and profiling.  sigma_out / sigma and z_all_out /
 * z_all use the tile-major workspace layout described above (buffers sized for ceil(R/32)*32 rays);
 * hl_render_fine takes z_all either tile-major (z_tiled = 1) or as caller rows (R,S) (z_tiled = 0). */
int hl_render_coarse(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                     const float *rays_o, const float *rays_d, const float *near, const float *far,
                     const float *z_vals, int64_t n_rays, int n_samples, float *sigma_out, void *stream);
int hl_render_importance(const float *sigma, const float *rays_d, const float *near, const float *far,
                         const float *z_vals, const float *u, int64_t n_rays, int n_samples, int n_importance,
                         float *z_all_out, void *stream);
/* The stages of the default (evaluate-once) schedule of hl_render_rays.  A "record" is the raw network output of one sample
 * point, float[4] = (sigma, r, g, b) before softplus / sigmoid; record and depth arrays are tile-major [ceil(R/32)][samples][32].
 *   hl_render_eval            full MLP at every given depth (z: NULL -> linspace, caller rows (R,S) with z_tiled = 0, or tile-major
 *                             with z_tiled = 1) -> records_out
 *   hl_render_importance_new  sample_pdf on the coarse records' densities (renderer.py:158-170, 533-563) -> the n_importance NEW
 *                             depths only, sorted (the reference sorts the union, :252-253; hl_render_composite merges instead)
 *   hl_render_composite       merge the coarse and the new depths and alpha-composite their records in depth order (:172-231) */
int hl_render_eval(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *rays_o,
                   const float *rays_d, const float *near, const float *far, const float *z, int z_tiled, int64_t n_rays,
                   int n_samples, float *records_out, void *stream);
/* hl_render_eval with the product mode of hl_render_rays: flags = 0 (fp32 MFMA), HL_RENDER_MLP_BF16X3 or HL_RENDER_MLP_FP16 */
int hl_render_eval_products(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *rays_o,
                            const float *rays_d, const float *near, const float *far, const float *z, int z_tiled, int64_t n_rays,
                            int n_samples, unsigned flags, float *records_out, void *stream);
int hl_render_importance_new(const float *records, const float *rays_d, const float *near, const float *far, const float *z_vals,
                             const float *u, int64_t n_rays, int n_samples, int n_importance, float *z_new_out, void *stream);
int hl_render_composite(const float *near, const float *far, const float *z_vals /* coarse rows or NULL */, const float *z_new,
                        const float *rec_coarse, const float *rec_new, int64_t n_rays, int n_samples, int n_importance,
                        unsigned flags, float *rgb, float *acc, float *depth, void *stream);
int hl_render_fine(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                   const float *rays_o, const float *rays_d, const float *near, const float *far,
                   const float *z_all /* or NULL -> linspace */, int z_tiled, int64_t n_rays, int n_total_samples,
                   unsigned flags, float *rgb, float *acc, float *depth, void *stream);

/* Training of the renderer: forward with saved activations + backward of the evaluate-once schedule, for
 * the tri-plane fitting loop (recon_NeRF/run_nerf_batch.py:236-265: render -> img/acc loss -> backward -> Adam) and any other loss
 * on rgb_map / acc_map of Renderer.render with test=False (human_diffusion/NeRF/renderer.py:172-281).  The importance depths are
 * drawn under no_grad in the reference (:243-253), so gradients flow through the evaluation of the 2N sample points and the
 * compositing only.  Matrices are "row = unit, column = sample point" (point order: the pass's tile-major record order at column
 * offset *_off), row stride *_stride floats, hl_render_train_rows() rows:
 *   activations (630 rows): [0,27) tri-plane features | [27,155) softplus(pts_linears.1) | [155,283) softplus(pts_linears.0) |
 *                           [283,411) softplus(pts_linears.2) | [411,539) feature_linear | [539,566) view encoding |
 *                           [566,630) softplus(views_linear)
 *   deltas (634 rows, dL/d pre-activation): [0,128) pts_linears.0 | [128,256) pts_linears.1 | [256,384) pts_linears.2 |
 *                           [384,512) feature_linear | [512,576) views_linear | 576 alpha_linear | [577,580) rgb_linear |
 *                           [580,607) dL/d tri-plane features | [607,634) the same, ray-major (written by hl_render_plane_grads*,
 *                           which therefore take `del` as scratch although it is declared const)
 * so that every weight gradient is one product delta_rows x activation_rows^T over the sample points (e.g. pts_linears.2.weight =
 * deltas[256:384] x activations[0:155]^T) and every bias gradient a row sum: hl_render_weight_grads.
 *   hl_render_composite_noise     hl_render_composite with `noise` (R, n_samples+n_importance) added to the raw density of sorted
 *                                 sample s of each ray (renderer.py:212, randn_like in training mode); NULL = none
 *   hl_render_eval_acts           hl_render_eval that also writes the activation matrix; round 5: on the fp16x2 kernel of the inference default
 *                                 (k_march_plw<2, ACTS>: fp32 products from two fp16 planes per operand, softplus in the log2 domain, activations stored in
 *                                 natural units) - 0.38 -> 0.20 ms per 2 048 rays x 128 samples; gradient tests unchanged
 *   hl_render_composite_backward  g_rgb (R,3), g_acc (R) -> d_records of both passes (float[4] = d/d(sigma, r, g, b) raw, record
 *                                 layout; zero on padding rays) and rows 576..579 of the delta matrix `del` (columns: coarse pass,
 *                                 then the new depths); scratch: hl_render_composite_backward_scratch_bytes()
 *   hl_render_mlp_backward        one pass's d_records + activations -> its columns of the delta matrix; round 5: the five transposed-weight products with fp16x2
 *                                 operands (weights as two fp16 planes behind the fp32 image of hl_render_mlp_pack_bwd; delta tiles split in registers, scaled per ray
 *                                 and sample by a power of two into fp16's range and scaled back exactly): 0.39 -> 0.27 ms per pass; gradient bounds unchanged
 *   hl_render_plane_grads         feature deltas of both passes -> d_planes (27,H,W), overwritten: the transposed bilinear lookup;
 *                                 each workgroup owns a tile of texels and accumulates in LDS (no global atomics) in 64-bit FIXED
 *                                 POINT whose unit comes from the largest |delta| of the call: integer additions commute, so the
 *                                 result is the same bits on every run (round 5; rounding per contribution 2^-43 of that maximum
 *                                 or finer); a delta that is not finite makes the output NaN.
 *                                 scratch: hl_render_plane_grads_scratch_bytes()
 *   hl_render_mlp_pack_bwd        transposed weights for hl_render_mlp_backward (redo after every optimizer step, like
 *                                 hl_render_mlp_pack)
 *   hl_render_weight_grads        all 14 parameter gradients from the two matrices over n_cols sample points (multiple of 32), ADDED
 *                                 to the tensors of `grads` (PyTorch layouts, zero them first); fp32 products from exact
 *                                 three-way bf16 splits of both operands, fp32 accumulation (k_wgrad).  The 256 point ranges
 *                                 write their partial results to `scratch` (hl_render_weight_grads_scratch_bytes(n_cols)) and
 *                                 k_wgrad_finish sums them in a fixed order: the same bits on every run (round 5; before: float
 *                                 atomics) */
/* Canonical-space training (use_canonical_space=True with test=False; README.md:123 TightCap fitting): the deformation has no
 * parameters and the points get no gradient, so the backward is the one above with two substitutions -
 *   hl_render_eval_points_acts     hl_render_eval_points that also writes the activation matrix (forward, per pass)
 *   hl_render_plane_grads_points   hl_render_plane_grads for sample points that are not on straight rays in tri-plane space:
 *                                  pts_coarse / pts_new are the canonical points hl_deform_rays wrote for the two passes, `bounds`
 *                                  is tp_input['t_world_bounds']; scratch: hl_render_plane_grads_points_scratch_bytes() */
int hl_render_eval_points_acts(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *pts_c,
                               const float *dirs_c, int64_t n_rays, int n_samples, float *records_out, float *act, int64_t act_stride,
                               int64_t act_off, void *stream);
size_t hl_render_plane_grads_points_scratch_bytes(int64_t n_rays, int n_samples, int n_importance);
int hl_render_plane_grads_points(int H, int W, const float *bounds, const float *pts_coarse, const float *pts_new, int64_t n_rays,
                                 int n_samples, int n_importance, const float *del, int64_t del_stride, float *d_planes, void *scratch,
                                 void *stream);
typedef struct hl_render_mlp_grads {
    float *pts0_w, *pts0_b, *pts1_w, *pts1_b, *pts2_w, *pts2_b, *feat_w, *feat_b, *alpha_w, *alpha_b, *views_w, *views_b, *rgb_w,
        *rgb_b;   /* same order and shapes as hl_render_mlp_params */
} hl_render_mlp_grads;
size_t hl_render_weight_grads_scratch_bytes(int64_t n_cols);
int hl_render_weight_grads(const float *del, int64_t del_stride, const float *act, int64_t act_stride, int64_t n_cols,
                           const hl_render_mlp_grads *grads, void *scratch, void *stream);
int hl_render_composite_noise(const float *near, const float *far, const float *z_vals, const float *z_new, const float *rec_coarse,
                              const float *rec_new, const float *noise, int64_t n_rays, int n_samples, int n_importance,
                              unsigned flags, float *rgb, float *acc, float *depth, void *stream);
size_t hl_render_mlp_bwd_packed_bytes(void);
int hl_render_mlp_pack_bwd(const hl_render_mlp_params *h_params, void *packed_bwd, void *stream);
void hl_render_train_rows(int *act_rows, int *del_rows);
int hl_render_eval_acts(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *rays_o,
                        const float *rays_d, const float *near, const float *far, const float *z, int z_tiled, int64_t n_rays,
                        int n_samples, float *records_out, float *act, int64_t act_stride, int64_t act_off, void *stream);
size_t hl_render_composite_backward_scratch_bytes(int64_t n_rays, int n_samples, int n_importance);
int hl_render_composite_backward(const float *near, const float *far, const float *z_vals, const float *z_new,
                                 const float *rec_coarse, const float *rec_new, const float *noise, const float *g_rgb,
                                 const float *g_acc, int64_t n_rays, int n_samples, int n_importance, unsigned flags,
                                 float *d_rec_coarse, float *d_rec_new, float *del, int64_t del_stride, void *scratch,
                                 void *stream);
int hl_render_mlp_backward(const void *mlp_packed, const void *mlp_bwd_packed, int H, int W, const float *bounds,
                           const float *rays_o, const float *rays_d, const float *near, const float *far, const float *z,
                           int z_tiled, int64_t n_rays, int n_samples, const float *d_records, const float *act,
                           int64_t act_stride, int64_t act_off, float *del, int64_t del_stride, int64_t del_off, void *stream);
int hl_render_plane_grads(int H, int W, const float *bounds, const float *rays_o, const float *rays_d, const float *near,
                          const float *far, const float *z_vals /* coarse rows or NULL */, const float *z_new,
                          int z_new_rows /* 1: rows (R, n_importance), faster; 0: tile-major as hl_render_importance_new wrote them */,
                          int64_t n_rays, int n_samples, int n_importance, const float *del, int64_t del_stride, float *d_planes,
                          void *scratch, void *stream);
size_t hl_render_plane_grads_scratch_bytes(int64_t n_rays);

/* Per-view ray generation on the device.  Replaces get_rays
 * (human_diffusion/SynBodyView_datasets.py:316-329), the float32 casts and the near=0 / far=1 fill of
 * sample_ray_batch (:422-433) and get_near_far (:370-403) for one pinhole camera: float64 arithmetic like the
 * reference's numpy, rounded to float32 where the reference casts.  h_Kinv = inv(K) (3,3), h_R (3,3) world->camera,
 * h_T (3), h_bounds (2,3) are HOST float64 arrays copied into the launch; outputs are device arrays of H*W rays in
 * row-major pixel order: rays_o, rays_d (R,3) (exact zeros of rays_d become 1e-8 as in the reference), near, far (R),
 * mask_at_box (R) bytes or NULL. */
int hl_camera_rays(const double *h_Kinv, const double *h_R, const double *h_T, const double *h_bounds, int H, int W,
                   float *rays_o, float *rays_d, float *near, float *far, unsigned char *mask_at_box, void *stream);

/* Canonical-space deformation of query points.  Replaces Renderer.deform_target2c /
 * deform_target2c_op (human_diffusion/NeRF/renderer.py:52-132, recon_NeRF/lib/renderer.py:60-140) including the external
 * pytorch3d knn_points(K=1): world -> SMPL space with h_R (3,3) / h_Th (3) (host float32), brute-force nearest body vertex among
 * verts_smpl4 (V,4: xyz of (vertices - Th) R, w ignored), then the per-vertex row of `table` (V,36: t[3] Rinv[9] pose_off[3]
 * shape_off[3] pose_off_big[3] Rbig[9] tbig[3] pad[3], built by humanliff_amd.NeRF.deform.deform_tables) applied in the
 * reference's order.  pts / dirs (P,3) device, dirs may be NULL; outputs can_pts, can_dirs (P,3), vertex_ids (P) or NULL. */
int hl_deform_points(const float *pts, const float *dirs, const float *h_R, const float *h_Th, const float *verts_smpl4,
                     const float *table, int n_vertices, int64_t n_points, float *can_pts, float *can_dirs, int *vertex_ids,
                     void *stream);
/* Rendering in canonical space (use_canonical_space=True, renderer.py:114-132, 192-201, 242-246): the stages and the driver.
 *   hl_deform_rays             sample points o + d*z of every ray (z as in hl_render_eval) and its unit direction, deformed like
 *                              hl_deform_points -> pts_c, dirs_c: float[4] per sample (xyz, w unused), tile-major [ceil(R/32)][S][32]
 *   hl_render_eval_points      hl_render_eval on given canonical points / directions; `bounds` is tp_input['t_world_bounds']
 *   hl_render_rays_canonical   evaluate-once schedule of hl_render_rays with the two stages above in front of each evaluate pass;
 *                              workspace: hl_render_canonical_workspace_bytes() */
int hl_deform_rays(const float *rays_o, const float *rays_d, const float *near, const float *far, const float *z, int z_tiled,
                   int64_t n_rays, int n_samples, const float *h_R, const float *h_Th, const float *verts_smpl4, const float *table,
                   int n_vertices, float *pts_c, float *dirs_c, void *scratch /* 8 bytes of device memory (work counter) */, void *stream);
int hl_render_eval_points(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds, const float *pts_c,
                          const float *dirs_c, int64_t n_rays, int n_samples, float *records_out, void *stream);
size_t hl_render_canonical_workspace_bytes(int64_t n_rays, int n_samples, int n_importance);
int hl_render_rays_canonical(const void *mlp_packed, const void *planes_packed, int H, int W, const float *t_bounds,
                             const float *rays_o, const float *rays_d, const float *near, const float *far, const float *z_vals,
                             const float *u, int64_t n_rays, int n_samples, int n_importance, unsigned flags, const float *h_R,
                             const float *h_Th, const float *verts_smpl4, const float *table, int n_vertices, float *rgb, float *acc,
                             float *depth, void *workspace, void *stream);

/* ------------------------------------------------------------------------
 * Path 1 — tri-plane UNet denoiser + Gaussian-diffusion sampler update
 * ------------------------------------------------------------------------ */

/* Architecture hyper-parameters, as UNetModel.__init__ receives them
 * (human_diffusion/improved_diffusion/unet.py:323-343, built by script_util.py:98-150).
 * Supported: dims=2, use_scale_shift_norm True or False, cond_type="controlnet", "AdaGN", "cross_attention", "concat" (a wider in_channels) or "", dropout=0,
 * conv_resample=True; use_3d_aware False (the shipped configuration) or True (not with AdaGN). */
typedef struct hl_unet_cfg {
    int in_channels, model_channels, out_channels, num_res_blocks;
    int n_levels;
    int channel_mult[8];
    int n_attention_ds;
    int attention_ds[8];        /* downsample rates at which attention is inserted */
    int num_heads, num_heads_upsample;
    int num_classes;            /* 0: not class conditional */
    int controlnet;             /* 1: cond_type == "controlnet" */
    int adagn;                  /* 1: cond_type == "AdaGN" (x_cond projected to one more summand of the timestep embedding) */
    int cross_attn;             /* 1: cond_type == "cross_attention": SpatialTransformer blocks (spatial_transformer.py, depth 1) in place of the
                                 * AttentionBlocks, attending to one context token per image = the AdaGN projection of x_cond */
    int aware3d;                /* 1: use_3d_aware: x, x_cond and the output are (B, 3*in_channels, H, W) tri-planes; the network runs on
                                 * the three planes side by side, (B, in_channels, H, 3W), and every ResBlock of the main towers feeds
                                 * each plane the axis means of the other two (unet.py:208-214, 566-570, 613-614) */
    int no_scale_shift;         /* 1: use_scale_shift_norm=False (unet.py:216-218): emb_layers emits C values that are ADDED to the first
                                 * convolution's output before out_layers' GroupNorm, instead of 2C values that scale and shift its result */
} hl_unet_cfg;

/* Bytes of the re-laid weights (conv OIHW -> GEMM-ready rows, emb_layers stacked). */
size_t hl_unet_packed_bytes(const hl_unet_cfg *cfg);

/* Bind a state_dict (the reference's key names) and pack it.
 * names/ptrs/numels: n_tensors parallel arrays; ptrs are device fp32 tensors in the
 * reference layouts (Conv2d OIHW, Conv1d (O,I,1), Linear (O,I)).  GroupNorm affine, biases
 * and label_emb are used in place (the caller keeps them alive); everything else is copied
 * into `packed`.  Replaces UNetModel.__init__ + load_state_dict for inference. */
int hl_unet_create(const hl_unet_cfg *cfg, int n_tensors, const char *const *names, const void *const *ptrs,
                   const int64_t *numels, void *packed, void *stream, void **handle);
void hl_unet_destroy(void *handle);

size_t hl_unet_workspace_bytes(void *handle, int B, int H, int W);

/* Replaces UNetModel.forward(x, timesteps, x_cond, y) (unet.py:550-615).
 * x, x_cond, out: (B, C, H, W) fp32 NCHW; t, y: (B) int64 (y NULL if not class conditional,
 * x_cond NULL only without controlnet).  t holds ORIGINAL-schedule integer timesteps
 * (respace.py:117-122 with rescale_timesteps=False); t_float, if non-NULL, overrides it with
 * fractional timesteps (rescale_timesteps=True). */
int hl_unet_forward(void *handle, const float *x, const int64_t *t, const float *t_float, const float *x_cond,
                    const int64_t *y, float *out, int B, int H, int W, void *workspace, void *stream);

/* The control encoder (input_blocks_cond) is independent of the main encoder up to the skip sums; by default
 * hl_unet_forward runs it on an internal second HIP stream, forked from and joined back into the caller's
 * stream with events (the low-resolution layers of the two branches then fill the chip together).
 * enable = 0 issues everything on the caller's stream. */
int hl_unet_set_overlap(void *handle, int enable);

/* Arithmetic of the large convolutions - all modes keep fp32 tensors and fp32 accumulators.
 * HL_CONV_FP32 (default): fp32-class products, fp32 accumulation.  3x3 / stride-1 layers take Winograd F(4x4,3x3) on v_mfma_f32_32x32x2_f32 (36 fp32
 *   multiplies per 4x4 outputs instead of 144; interpolation points 0, +-3/4, +-3/2, inf) where 32x16-pixel x 32-channel
 *   workgroups fill the chip (the 256- and 128-pixel levels at batch 4), Winograd F(2x2,3x3) (16 per 2x2 instead of 36) on the
 *   smaller levels, everything else the direct implicit GEMM.  Round 5: every 3x3 / stride-1 layer (also behind the nearest-x2 upsample) and every
 *   1x1 / stride-1 layer with Cout a multiple of 192, Cin a multiple of 32 / 96 and enough work (3x3: from 8 workgroups' worth of 256 pixels x 192
 *   channels, split-K below 100; 1x1: from 12) is a DIRECT convolution whose fp32 products come from TWO fp16 planes per operand on
 *   v_mfma_f32_32x32x16_f16 (k_conv_h2s, k_conv1_h2s: 8x16-pixel / 128-pixel tiles, two workgroups per CU): activation x = h0 + h1 (h0 = the nearest
 *   fp16, h1 = the nearest fp16 of the residual: |x - h0 - h1| <= 2^-23 |x| while both planes are normal), weight planes nearest even,
 *   h1 w0 + h0 w1 + h0 w0 accumulated in fp32.  Round 6 - the planes are SCALE-INVARIANT: every output channel's weights are multiplied by the power of
 *   two that puts the channel's largest |w| into [2^13, 2^14) before the split (2^-22 of it whatever the magnitude: zero_module convolutions at 1e-4,
 *   unet.py:149, lose nothing), and the staged input by the power of two sx with |x| sx <= 32752 - from sqrt(sum x^2) of the tensor (the group totals
 *   its producers left; the single-op entry points form them) for a raw input, from max|gamma'| sqrt(n_group) + max|beta'| behind a fused GroupNorm -
 *   so no plane overflows or goes subnormal where the magnitude is known; both scales leave in the epilogue (powers of two: exact).  Unknown magnitude
 *   (a GroupNorm given as coefficient arrays, the attention output in front of proj_out): sx = 1, and a value beyond fp16's range becomes inf / NaN
 *   (round 5 saturated silently).  Error of the fp32 direct kernel's class at EVERY scale, checked against float64 with weights x 2^0 ... 2^-18 and
 *   activations x 2^-10 ... 2^10 (tests/test_unet_gpu.py::test_conv_fp16x2_products_are_scale_invariant: rel-L2 2.2e-7 ... 6.4e-7 against 2.0e-7 ...
 *   7.3e-7 of HL_CONV_FP32_DIRECT, bit-identical across the scales).  A GroupNorm (+ SiLU) in front of such a layer is applied
 *   while the kernel stages its input (no pass over the tensor).  The F(4x4,3x3) / F(2x2,3x3) kernels keep the layers the direct kernels do not
 *   take (the 27-channel input convolution, Cout not a multiple of 192, small single-op calls).
 * HL_CONV_FP32_MFMA: HL_CONV_FP32 without those two kernels - every product on v_mfma_f32_32x32x2_f32 (the default of rounds 3-4).
 * HL_CONV_FP32_F23: the same without F(4x4,3x3) (the arithmetic of the round-2 library).
 * HL_CONV_FP32_DIRECT: the direct implicit GEMM only - every product of the reference's sum is formed exactly once.
 * HL_CONV_BF16X3 (opt-in, direct only): each product a*b is formed on the bf16 matrix pipe from exact three-way splits
 *   a = ah+am+al, b = bh+bm+bl (8 significand bits per bf16 plane) as ah*bh + ah*bm + am*bh + ah*bl + am*bm + al*bh; the three
 *   dropped terms are <= 3*2^-24 |a*b|, the size of one fp32 rounding.
 * The modes are not bit-identical to each other; all meet the same parity bounds (tests/test_unet_gpu.py,
 * tests/test_fullsize_gpu.py: production UNet vs the CPU oracle about 5e-6 max-abs on O(0.5) outputs in every mode; against the direct mode the F(4x4) mode differs
 * by 7.9e-6 max-abs / 1.4e-6 rms, the F(2x2) mode by 6.4e-6 / 1.0e-6).
 * Only layers with Cout a multiple of 96 (the DMA tile) are affected. */
#define HL_CONV_FP32 0
#define HL_CONV_BF16X3 1
#define HL_CONV_FP32_DIRECT 2
#define HL_CONV_FP32_F23 3
/* HL_CONV_BF16 (opt-in, direct only): 16-bit MFMA arithmetic for the TRAINING path - what the reference's train scripts select with
 *   --use_amp True (train_util.py:214 autocast): every activation is rounded to bf16 (nearest even) and multiplied with the weight's two
 *   leading bf16 planes (16 significand bits), fp32 accumulation on v_mfma_f32_32x32x16_bf16; tensors stay fp32 in HBM, master weights
 *   fp32.  Relative error per product <= 2^-9; NOT an inference default.  Forward (hl_conv2d_nhwc_mode) and backward-data
 *   (hl_conv2d_nhwc_bwd_data) take it; weight gradients stay fp32. */
#define HL_CONV_BF16 4
/* HL_CONV_FP16 (opt-in): fp16 operands / fp32 accumulation (v_mfma_f32_32x32x16_f16) on the 3x3 / stride-1 layers (k_conv_h16, also behind a nearest-x2 upsample) and the 1x1 layers (k_conv1_h16) - the
 * operand precision of the reference's own convolutions on its hardware (TF32: 10 explicit significand bits) and of its autocast training;
 * every other layer as HL_CONV_FP32.  HL_CONV_BF16 takes the same kernel with bf16 operands on those layers. */
#define HL_CONV_FP16 5
#define HL_CONV_FP32_MFMA 6
int hl_unet_set_conv_mode(void *handle, int mode);

/* Instrumentation for the roofline measurement (bench.py): with profiling enabled every kernel launch
 * of hl_unet_forward is bracketed by HIP events on the caller's stream.  hl_unet_profile_read waits for
 * them and returns, per category {0 conv/GEMM, 1 GroupNorm, 2 attention, 3 embeddings+prep}, the summed
 * kernel time (ms), the algorithmic FLOPs (2*MAC) and the launch count since the last read (host arrays
 * of 4).  Not meant for the timed production path. */
int hl_unet_profile(void *handle, int enable);
int hl_unet_profile_read(void *handle, double *h_ms, double *h_flops, int64_t *h_launches);
/* same, plus the FLOPs the kernels actually issued to the matrix pipe (h_exec_flops, may be NULL): Winograd layers issue
 * 16/36 of their algorithmic multiplies, HL_CONV_BF16X3 layers six bf16 products per fp32 product. */
int hl_unet_profile_read_ex(void *handle, double *h_ms, double *h_flops, double *h_exec_flops, int64_t *h_launches);

/* The convolution shape that took the most time in the spans of the last hl_unet_profile_read(_ex): h_vals = {total ms over its launches,
 * algorithmic FLOPs per launch, FLOPs issued to the matrix pipe per launch, number of launches} (averages over the launches of the shape),
 * h_key = {kernel family (as in hl_unet_dispatch_census), resolution level, 1 if behind a nearest-x2 upsample, Cout, kernel size} - the grouping
 * of a rocprofv3 per-kernel, per-grid row (layers that differ only in the input channel count share it).  The time is the convolution kernel's own (an event behind the
 * GroupNorm pre-pass k_gn_apply(_blk) separates the two; a split-K finish pass, where there is one, is included). */
int hl_unet_profile_dominant(void *handle, double *h_vals, int *h_key);

/* Which kernel family every convolution of the LAST hl_unet_forward took: h_counts[path * 8 + level] launches, path 0 = direct
 * implicit GEMM (k_conv_dma / k_conv), 1 = Winograd F(2x2,3x3) (k_conv_wino), 2 = bf16x3 emulation (k_conv_bf3), 3 = Winograd F(4x4,3x3)
 * (k_conv_wino4); level = log2(H / H_out) of the layer's output.  Kernel selection depends on the batch size (a layer takes a Winograd
 * kernel only where its workgroups fill the chip), so parity tests use this to state WHICH dispatch they covered. */
int hl_unet_dispatch_census(void *handle, int64_t *h_counts);
/* The same with `rows` <= 5 rows of 8 levels: direct | Winograd F(2x2) | bf16x3 / 16-bit operand kernels | Winograd F(4x4) | the direct convolutions of the
 * default mode with fp16x2 products (k_conv_h2s, k_conv1_h2s: two fp16 planes per operand, three partial products, fp32 accumulation). */
int hl_unet_dispatch_census_ex(void *handle, int64_t *h_counts, int rows);

/* Fused sampler update (everything after the model call in p_sample / ddim_sample,
 * gaussian_diffusion.py:293-333, 356-388, 484-529) for EPSILON prediction with a fixed
 * variance.  coef: (T, 8) fp32 per-kept-timestep table built by the host mirror from the
 * float64 schedule: [sqrt_recip_acp, sqrt_recipm1_acp, c0, c1, c2, 1/coef1, coef2/coef1, 0] (coef1/2 = posterior_mean_coef1/2) where
 *   mode 0 (p_sample):  x0 = clip(r*x - rm1*eps); sample = (c0*x0 + c1*x) + (t!=0) * c2 * noise
 *   mode 1 (ddim):      x0 = clip(r*x - rm1*eps); e = (r*x - x0)/rm1;
 *                       sample = (x0*c0 + c1*e) + (t!=0) * c2 * noise
 *   mode 2 / 3:         as 0 / 1 with `eps` holding the already processed pred_xstart (the caller applied denoised_fn and the
 *                       clamp, gaussian_diffusion.py:293-299); `clip` is ignored
 *   mode 4 / 5:         x_{t-1} prediction (ModelMeanType.PREVIOUS_X, :300-304, 335-343): `eps` holds the model's x_{t-1}; x0 = clip(eps/coef1 -
 *                       coef2/coef1 * x); mode 4: sample = eps + noise term (the prediction is the mean); mode 5: the ddim update from x0
 *   log_variance:       NULL = fixed variance (c2 of the table); else the per-element model_log_variance of a learned-variance model
 *                       (gaussian_diffusion.py:262-276), same shape as x: the noise term of mode 0 / 2 becomes (t!=0)*exp(lv/2)*noise
 * t: (B) int64 indices into the table of T rows (a t outside [0, T) reads nothing out of bounds and turns that
 * sample's outputs into NaN; the reference raises IndexError, which the Python binding reproduces); n_per_sample = C*H*W; noise may be NULL (no noise term:
 * `sample` is then the model mean of p_mean_variance); pred_xstart may be NULL. */
int hl_diffusion_step(int mode, const float *x, const float *eps, const float *noise, const float *coef,
                      const int64_t *t, float *sample, float *pred_xstart, int64_t n_per_sample, int B, int T, int clip,
                      const float *log_variance, void *stream);

/* Single ops of the UNet path, exposed for parity tests and profiling (NHWC fp32). */
int hl_conv2d_nhwc(const float *in, int N, int H, int W, int Cin, const float *w_oihw, const float *bias, int Cout,
                   int ks, int stride, int upsample, const float *coefA, const float *coefB, int silu,
                   const float *residual, float *out, void *scratch, size_t scratch_bytes, void *stream);
/* hl_conv2d_nhwc with an explicit arithmetic mode (HL_CONV_*); the scratch also holds the mode's second weight layout:
 * scratch >= 4*Cout_pad*K (rounded up to 256) + extra, Cout_pad = Cout rounded up to 64, K = Cin*ks*ks; extra = 6*Cout_pad*K bytes
 * (HL_CONV_BF16X3), 64*Cout*Cin (HL_CONV_FP32_F23, 3x3 layers) or 144*Cout*Cin (HL_CONV_FP32, 3x3 layers); what is left beyond
 * that serves the materialised GroupNorm input (N*H*W*Cin floats, when coefA is given) and split-K partial sums. */
int hl_conv2d_nhwc_mode(int conv_mode, const float *in, int N, int H, int W, int Cin, const float *w_oihw, const float *bias,
                        int Cout, int ks, int stride, int upsample, const float *coefA, const float *coefB, int silu,
                        const float *residual, float *out, void *scratch, size_t scratch_bytes, void *stream);
/* ---- backward of the UNet (gaussian_diffusion.py:688-772 -> loss.backward() through unet.py:550-615) --------
 * Backward-DATA of a convolution is a forward convolution of the output gradient with the flipped, channel-transposed weights
 * (hl_conv2d_nhwc_bwd_data); stride 2 goes through hl_zero_stuff2_nhwc first, the nearest-x2 upsample through
 * hl_upsample2_backward_nhwc afterwards.  humanliff_amd/improved_diffusion/unet_train.py holds the autograd.Functions.
 *
 * Weight gradient: dW[co][ci][ky][kx] = sum_p dY[p][co] * X[p*stride + (ky,kx) - pad][ci] and db[co] = sum_p dY[p][co] with dw (Cout, Cin,
 * ks, ks) in the reference's OIHW parameter layout; x (N,H,W,Cx), dy (N,Hout,Wout,Cy) dense NHWC with Cx >= Cin, Cy >= Cout (zero-padded
 * channels are ignored); `upsample`: the convolution ran on the nearest-x2 upsampled x.  (hl_conv2d_wgrad_nhwc_ws below.) */
/* hl_conv2d_nhwc_bwd_data: d input of a convolution from d output, through the forward kernels: dy (N,Ho,Wo,Cy) dense NHWC with
 * Cy >= Cout a multiple of 16 (zero-padded channels), w the convolution's own (Cout, Cin, ks, ks) weights - read flipped and
 * channel-transposed while they are re-laid for the kernel, no flipped copy is made -, dx (N,H,W,Cx) with Cx >= Cin (channels
 * [Cin, Cx) are not written).  stride 2 (ks 3): H = 2*Ho; upsample: the convolution ran on the nearest-x2 image, (Ho,Wo) = (2H,2W).
 * scratch: the re-laid weights (5 * round_up(Cin,64) * Cy * ks^2 floats) + 16 MiB + the zero-stuffed / upsampled gradient. */
int hl_conv2d_nhwc_bwd_data(int conv_mode, const float *dy, int N, int Ho, int Wo, int Cy, const float *w_oihw, int Cout, int Cin, int ks,
                            int stride, int upsample, float *dx, int Cx, void *scratch, size_t scratch_bytes, void *stream);
/* hl_conv2d_wgrad_nhwc_ws: the weight / bias gradients, without atomics.  3x3 layers: k_conv_wgrad_t (a workgroup owns a 64x64 channel block of
 * dW for all nine taps; dY rows and input patch of an 8x8-pixel tile staged in LDS once); 1x1 layers: k_conv_wgrad_1x1 (192 x 64 channel
 * block, 64-pixel tiles).  Per-slab partial blocks go to `scratch` and k_wgrad_finish sums them in a fixed order: dw / db are plainly
 * stored (no need to zero them) and bit-reproducible.  Needs Cx, Cy multiples of 4 and `scratch` of
 * hl_conv2d_wgrad_scratch_bytes(...) bytes (0 = this geometry has no kernel: channel counts that are not multiples of 4, or a 1x1
 * convolution with stride 2 / behind an upsample - hl_conv2d_wgrad_nhwc_ws refuses those; the atomics-based entry of rounds 2-3 is gone). */
size_t hl_conv2d_wgrad_scratch_bytes(int N, int H, int W, int Cx, int Cy, int ks, int stride, int upsample, int Cout, int Cin);
int hl_conv2d_wgrad_nhwc_ws(const float *x, int N, int H, int W, int Cx, const float *dy, int Cy, int ks, int stride, int upsample,
                            float *dw, int Cout, int Cin, float *db, void *scratch, size_t scratch_bytes, void *stream);
/* The same with an arithmetic mode: HL_CONV_FP16 / HL_CONV_BF16 round both operands to 16 bits and accumulate in fp32
 * (v_mfma_f32_32x32x16, k_conv_wgrad_h16) on the 3x3 / stride-1 layers; every other layer and mode as hl_conv2d_wgrad_nhwc_ws. */
int hl_conv2d_wgrad_nhwc_ws_mode(int conv_mode, const float *x, int N, int H, int W, int Cx, const float *dy, int Cy, int ks, int stride,
                                 int upsample, float *dw, int Cout, int Cin, float *db, void *scratch, size_t scratch_bytes, void *stream);
/* y (N,HW,C dense) = silu ? silu(x*A + B) : x*A + B with the per-(n,c) affine of hl_groupnorm_coef; x has a channel pitch. */
int hl_gn_apply_nhwc(const float *x, long x_pitch, int N, int HW, int C, const float *coefA, const float *coefB, int silu, float *y,
                     void *stream);
/* backward of y = silu?(x*A + B): with du = dout * silu'(x*A + B) (or dout), S[n][c] = (sum_p du, sum_p du*x), written (N,C,2);
 * per-workgroup partial sums in the caller's scratch (hl_gn_backward_scratch_bytes) added in a fixed order - no atomics, the same
 * bits on every run; then dx = k1[n,c]*du + k2[n,c]*x + k3[n,c] (+ dx_add), the (N,C) coefficients being the caller's small
 * tensor algebra on S, the group statistics and the affine parameters (unet_train.py). */
size_t hl_gn_backward_scratch_bytes(int N, int HW, int C);
int hl_gn_backward_reduce(const float *x, long x_pitch, const float *dout, int N, int HW, int C, const float *coefA, const float *coefB,
                          int silu, float *S, void *scratch, size_t scratch_bytes, void *stream);
int hl_gn_backward_apply(const float *x, long x_pitch, const float *dout, int N, int HW, int C, const float *coefA, const float *coefB,
                         int silu, const float *k1, const float *k2, const float *k3, const float *dx_add, float *dx, void *stream);
/* The two as the training path uses them (unet_train.py), the (N,C) algebra between the passes in a kernel of its own:
 * forward   y = silu?( GroupNorm32(x) [* (1 + scale) + shift] ), x / y dense (N,H,W,C); scale_shift (N,2C) = [scale | shift] or NULL
 *           (unet.py:203-206); also returns the affine coefA / coefB (N,C) and gstat (N,32,2) = (mean, rstd) per group for the backward.
 *           scratch: N * 32 KiB.
 * backward  dx (may be NULL), dgamma / dbeta (C) (written, not accumulated), dscale_shift (N,2C) (iff scale_shift); scratch: N*C*5 floats
 *           + hl_gn_backward_scratch_bytes(N, H*W, C).  Deterministic (fixed-order sums). */
int hl_groupnorm_train_forward(const float *x, int N, int H, int W, int C, const float *gamma, const float *beta, const float *scale_shift,
                               int silu, float *coefA, float *coefB, float *gstat, float *y, void *scratch, size_t scratch_bytes, void *stream);
int hl_groupnorm_train_backward(const float *x, const float *dout, int N, int H, int W, int C, const float *coefA, const float *coefB, int silu,
                                const float *gstat, const float *gamma, const float *beta, const float *scale_shift, float *dx,
                                float *dgamma, float *dbeta, float *dscale_shift, void *scratch, size_t scratch_bytes, void *stream);
/* dx (N,H,W,C) = sums of the 2x2 blocks of d_up (N,2H,2W,C): backward of the nearest-x2 upsample (unet.py:77). */
int hl_upsample2_backward_nhwc(const float *d_up, int N, int H, int W, int C, float *dx, void *stream);
/* z (N,2Ho,2Wo,C): z[2y][2x] = dy[y][x], zero elsewhere - backward-data of a stride-2 3x3 conv = flipped 3x3 conv of z. */
int hl_zero_stuff2_nhwc(const float *dy, int N, int Ho, int Wo, int C, float *z, void *stream);

/* hl_conv2d_nhwc_mode followed by the GroupNorm32 affine of its OUTPUT (nn.py:17-19,100: y = out*A[n,c] + B[n,c] for the layer
 * that normalises `out` next): the statistics come from the epilogue of the kernel that stored `out` (fixed-point group totals added
 * with integer atomics: order-free, bit-reproducible) - the tensor is not read again; *h_used_stats returns nonzero when the epilogue
 * emitted them, 0 when the kernel
 * path taken emits none and the statistics were computed from the tensor instead.  scratch as hl_conv2d_nhwc_mode plus
 * (N*Hout*Wout/32 + 1)*Cout*8 + N*32 KiB bytes. */
int hl_conv2d_nhwc_gn(int conv_mode, const float *in, int N, int H, int W, int Cin, const float *w_oihw, const float *bias, int Cout,
                      int ks, int stride, int upsample, const float *coefA, const float *coefB, int silu, const float *residual,
                      float *out, const float *gamma, const float *beta, float *next_coefA, float *next_coefB, int *h_used_stats,
                      void *scratch, size_t scratch_bytes, void *stream);
int hl_groupnorm_coef(const float *x, int N, int H, int W, int C, const float *gamma, const float *beta,
                      const float *emb /* (N,2C) or NULL */, float *coefA, float *coefB, void *scratch,
                      size_t scratch_bytes, void *stream);
int hl_attention_nhwc(const float *qkv, int N, int T, int C, int heads, float *out, void *stream);
/* The attention as hl_unet_forward runs it in `conv_mode`: HL_CONV_FP32 (the default) forms both products (scores, probabilities x values) from fp16x2 operands on
 * v_mfma_f32_32x32x16_f16 where the kernel has that form (head sizes 96 and 192 on short sequences: the UNet's 32-, 16- and 8-pixel levels); every other mode = hl_attention_nhwc
 * (fp32 MFMA).  Round 6: the V tiles are staged under a running power-of-two scale, so the result follows any scaling of V exactly (2^-14 ... 2^10 tested bit for bit); Q and K
 * are split as they are (their error enters the scores absolutely and the softmax flattens it). */
int hl_attention_nhwc_mode(int conv_mode, const float *qkv, int N, int T, int C, int heads, float *out, void *stream);
/* Backward of hl_attention_nhwc (QKVAttention, unet.py:255-274, under train_util.py:200-246): qkv as in the forward, out = the forward's
 * output (N,T,C), dout = its gradient -> dqkv (N,T,3C).  fp32 MFMA, probabilities recomputed (flash-style), every output summed by one wave in
 * a fixed order (deterministic).  scratch: hl_attention_backward_scratch_bytes() (per query: row maximum, 1/row sum, dO.O; head sizes that are
 * not a multiple of 32 materialise P and dS there instead). */
size_t hl_attention_backward_scratch_bytes(int N, int T, int C, int heads);
int hl_attention_nhwc_backward(const float *qkv, const float *out, const float *dout, int N, int T, int C, int heads, float *dqkv,
                               void *scratch, size_t scratch_bytes, void *stream);
/* timestep_embedding (nn.py:103-121): t int64 (B) or t_float fp32 (B) -> out (B, dim), dim even */
int hl_timestep_embedding(const int64_t *t, const float *t_float, int B, int dim, float *out, void *stream);

/* Developer / test switch: the number of workgroups from which the convolution dispatch takes k_conv_h16 in the 16-bit modes (default 48,
 * or HL_H16_MIN_BLOCKS read once at the first launch); v < 0 restores the default.  The unit tests run the kernel on single tiles with it. */
int hl_debug_set_h16_min_blocks(long v);
/* test switch of the single-convolution entry points (hl_conv2d_nhwc*): where the power-of-two scale of a raw input of the fp16x2 kernels comes from.
   0 (default): an exact abs-max pass over the input (tensor_absmax); 1: the fixed-point group totals (sum x^2) - what the producers' epilogues leave inside
   hl_unet_forward - formed here by one pass (tensor_totals), so that the network's scale source can be tested on single layers. */
int hl_debug_set_single_op_scale_source(int from_totals);

/* sample_pdf's uniforms on the device, bit for bit (replaces `u = torch.rand(...)` on the CPU generator + upload, NeRF/renderer.py:545):
 * continues the mt19937 stream of ATen's CPU generator from `state` (624 words, device) at position `pos` (words of the current block
 * already drawn; 624 = block used up, also the freshly seeded generator) and writes the next n floats u = (word & (2^24 - 1)) * 2^-24 to
 * out; state_out (625 words, device) receives the state and position behind the last number - the host side advances its generator
 * with it (humanliff_amd/NeRF/cpu_rng.py).  One workgroup; enqueue-only. */
int hl_mt19937_uniform(const uint32_t *state, int pos, float *out, int64_t n, uint32_t *state_out, void *stream);
/* test switch: the largest number of words one walk launch of hl_mt19937_uniform produces (default 2^28: the walk addresses its output through a 32-bit buffer
 * range; longer requests continue from the state the piece before left).  words <= 0 restores the default.  The unit tests run the chaining with small pieces. */
int hl_debug_set_mt19937_piece(int64_t words);

#ifdef __cplusplus
}
#endif
#endif /* HUMANLIFF_HIP_H */
