// Fused Gaussian-diffusion sampler update (one launch per denoise step), MI355X.
//
// Replaces the ~15 elementwise PyTorch kernels after the model call in
//   GaussianDiffusion.p_sample      human_diffusion/improved_diffusion/gaussian_diffusion.py:356-388
//   GaussianDiffusion.ddim_sample   gaussian_diffusion.py:484-529
// (through p_mean_variance :293-326, _predict_xstart_from_eps :328-333, q_posterior_mean_variance :208-230,
// _predict_eps_from_xstart :345-349) for EPSILON prediction with a fixed variance.  The per-timestep scalars
// come from a device-resident (T,8) table instead of _extract_into_tensor's per-call H2D copies (:850-863).
// Arithmetic follows the reference's fp32 op order (file built with -ffp-contract=off), so the result is
// bit-identical to the eager PyTorch expression on the same inputs.
#include "hl_common.h"

namespace {

template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void k_step(const float *__restrict__ x, const float *__restrict__ eps,
                                              const float *__restrict__ noise, const float *__restrict__ coef,
                                              const int64_t *__restrict__ t, float *__restrict__ sample,
                                              float *__restrict__ x0_out, long n, int T, int clip, int has_noise, int x0_given,
                                              const float *__restrict__ logvar, int xprev_given) {
    const int b = blockIdx.y;
    const int64_t tb = t[b];
    // a timestep outside the (T,8) table (the reference's numpy indexing raises IndexError, gaussian_diffusion.py:859) never
    // reads out of bounds: row 0 is read and every output of this sample becomes NaN
    const bool in_range = tb >= 0 && tb < (int64_t)T;
    const float *c = coef + (in_range ? tb : 0) * 8;
    const float poison = in_range ? 0.f : __builtin_nanf("");
    const float r = c[0] + poison, rm1 = c[1], c0 = c[2], c1 = c[3], q0 = c[5], q1 = c[6];
    const float nz = has_noise ? (tb != 0 ? 1.f : 0.f) * c[4] : 0.f;
    const long base = (long)b * n;
    // logvar (learned variances, gaussian_diffusion.py:262-276): per-element model_log_variance replaces the table's sigma
    auto one = [&](float xv, float ev, float nv, float lv, float &sv, float &x0v) {
        // x0_given: `eps` holds pred_xstart already processed by the caller (denoised_fn + clamp, gaussian_diffusion.py:293-299)
        // xprev_given: `eps` holds the model's x_{t-1} prediction (gaussian_diffusion.py:300-304): x0 = xprev / coef1 - coef2 / coef1 * x_t
        // (:335-343) and the model mean of the ancestral step is the prediction itself
        float x0 = x0_given ? ev + poison : (xprev_given ? q0 * ev - q1 * xv + poison : r * xv - rm1 * ev);
        if (clip && !x0_given) x0 = fminf(fmaxf(x0, -1.f), 1.f);
        float mean;
        if (MODE == 0) {
            mean = xprev_given ? ev + poison : c0 * x0 + c1 * xv;
        } else {
            const float e2 = (r * xv - x0) / rm1;
            mean = x0 * c0 + c1 * e2;
        }
        const float sg = logvar ? (has_noise && tb != 0 ? expf(0.5f * lv) : 0.f) : nz;
        sv = mean + sg * nv;
        x0v = x0;
    };
    if (VEC) {
        const long n4 = n >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            const f32x4 xv = reinterpret_cast<const f32x4 *>(x + base)[i];
            const f32x4 ev = reinterpret_cast<const f32x4 *>(eps + base)[i];
            const f32x4 nv = reinterpret_cast<const f32x4 *>(noise + base)[i];
            const f32x4 lv = logvar ? reinterpret_cast<const f32x4 *>(logvar + base)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 sv, zv;
#pragma unroll
            for (int k = 0; k < 4; ++k) { float s, z; one(xv[k], ev[k], nv[k], lv[k], s, z); sv[k] = s; zv[k] = z; }
            reinterpret_cast<f32x4 *>(sample + base)[i] = sv;
            if (x0_out) reinterpret_cast<f32x4 *>(x0_out + base)[i] = zv;
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
            float s, z;
            one(x[base + i], eps[base + i], noise[base + i], logvar ? logvar[base + i] : 0.f, s, z);
            sample[base + i] = s;
            if (x0_out) x0_out[base + i] = z;
        }
    }
}

}  // namespace

extern "C" int hl_diffusion_step(int mode, const float *x, const float *eps, const float *noise, const float *coef,
                                 const int64_t *t, float *sample, float *pred_xstart, int64_t n_per_sample, int B, int T, int clip,
                                 const float *log_variance, void *stream) {
    HL_REQUIRE(x && eps && coef && t && sample, "hl_diffusion_step: null argument");
    const int has_noise = noise != nullptr;
    if (!noise) noise = x;  // never contributes (factor 0); keeps the loads in bounds
    HL_REQUIRE(mode >= 0 && mode <= 5, "hl_diffusion_step: mode %d", mode);
    const int x0_given = (mode >> 1) & 1, xprev_given = mode >> 2;
    mode &= 1;
    HL_REQUIRE(n_per_sample > 0 && B > 0 && T > 0, "hl_diffusion_step: bad sizes");
    const bool vec = (n_per_sample % 4 == 0) && (((uintptr_t)x | (uintptr_t)eps | (uintptr_t)noise | (uintptr_t)sample |
                                                   (uintptr_t)pred_xstart | (uintptr_t)log_variance) % 16 == 0);
    const long work = vec ? n_per_sample / 4 : n_per_sample;
    long gx = (work + 255) / 256;
    if (gx > 1024) gx = 1024;
    dim3 grid((unsigned)gx, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
#define HL_GO(M, V) hipLaunchKernelGGL((k_step<M, V>), grid, dim3(256), 0, st, x, eps, noise, coef, t, sample, pred_xstart, (long)n_per_sample, T, clip, has_noise, x0_given, log_variance, xprev_given)
    if (mode == 0) { if (vec) HL_GO(0, true); else HL_GO(0, false); }
    else { if (vec) HL_GO(1, true); else HL_GO(1, false); }
#undef HL_GO
    return hl::check_launch("k_step");
}
