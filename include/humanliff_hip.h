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
#define HL_RENDER_NORMALIZE_DEPTH 2u /* renderer.py:272-274 (human_diffusion twin only) */

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
 *   sigma_coarse_out (R,n_samples) / z_all_out (R,n_samples+n_importance): optional
 *            copies of the intermediates (NULL to skip), for parity tests.
 */
int hl_render_rays(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                   const float *rays_o, const float *rays_d, const float *near, const float *far,
                   const float *z_vals, const float *u, int64_t n_rays, int n_samples, int n_importance,
                   unsigned flags, float *rgb, float *acc, float *depth, void *workspace, void *stream);

/* The three stages of hl_render_rays, exposed for tests and profiling. */
int hl_render_coarse(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                     const float *rays_o, const float *rays_d, const float *near, const float *far,
                     const float *z_vals, int64_t n_rays, int n_samples, float *sigma_out, void *stream);
int hl_render_importance(const float *sigma, const float *rays_d, const float *near, const float *far,
                         const float *z_vals, const float *u, int64_t n_rays, int n_samples, int n_importance,
                         float *z_all_out, void *stream);
int hl_render_fine(const void *mlp_packed, const void *planes_packed, int H, int W, const float *bounds,
                   const float *rays_o, const float *rays_d, const float *near, const float *far,
                   const float *z_all /* (R,S) or NULL -> linspace */, int64_t n_rays, int n_total_samples,
                   unsigned flags, float *rgb, float *acc, float *depth, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HUMANLIFF_HIP_H */
