"""ORACLE (test infrastructure, not product code) — CPU restatement of the
tri-plane NeRF volume renderer.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (humanliff_amd) never does.

Pinned against golden vectors generated from the reference itself
(tests/golden/gen_golden.py -> tests/golden/render_*.npz, checked in
tests/test_oracle_render.py).

Each function restates, with plain fp32 torch ops on the CPU, what the
reference computes (citations are into /root/reference/):

  plane_features   human_diffusion/NeRF/renderer.py:486-531 (project_onto_planes,
                   sample_from_planes; F.grid_sample bilinear / zeros /
                   align_corners=False)
  view_encoding    human_diffusion/NeRF/fields.py:45-85
  mlp              human_diffusion/NeRF/renderer.py:134-156
  importance_z     human_diffusion/NeRF/renderer.py:158-170, 533-563, 252-253
  composite        human_diffusion/NeRF/renderer.py:172-231
  render_rays      human_diffusion/NeRF/renderer.py:234-281 and the chunk body of
                   human_diffusion/scripts/triplane_sample_layered.py:262-279
  render_rays_recon  recon_NeRF/lib/renderer.py:244-295 (the fitting twin: module-owned tri-planes gathered by
                   (instance_idx, cloth_layer_index), unclamped depth); pinned by tests/golden/recon_twin.npz
"""
import math

import torch
import torch.nn.functional as F


def _bilinear_zeros(img, gx, gy):
    """img (C,H,W); gx,gy (M,) normalised coords -> (M,C).

    grid_sample(align_corners=False, padding_mode='zeros'): pixel = ((g+1)*size-1)/2,
    the four neighbours weighted by the opposite-corner areas, taps outside the
    image contribute zero.
    """
    C, H, W = img.shape
    ix = ((gx + 1) * W - 1) / 2
    iy = ((gy + 1) * H - 1) / 2
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    flat = img.reshape(C, H * W)

    def tap(xi, yi, w):
        ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
        xi = xi.clamp(0, W - 1).long()
        yi = yi.clamp(0, H - 1).long()
        v = flat[:, yi * W + xi]  # (C,M)
        return (v * (w * ok.to(w.dtype))[None]).t()

    return tap(x0, y0, w_nw) + tap(x1, y0, w_ne) + tap(x0, y1, w_sw) + tap(x1, y1, w_se)


def plane_features(planes, pts, bounds):
    """planes (3,9,H,W), pts (M,3), bounds (2,3) -> (M,27).

    Feature c = plane*9 + group*3 + k.  Plane 0 looks up (x,y), plane 1 (x,z),
    plane 2 (z,y); group 1 shifts the first grid coordinate by 1/H, group 2 the
    second (renderer.py:521-526 - both use 1/H).
    """
    H = planes.shape[-2]
    n = 2 * (pts - bounds[0:1]) / (bounds[1:2] - bounds[0:1]) - 1
    x, y, z = n[:, 0], n[:, 1], n[:, 2]
    uv = [(x, y), (x, z), (z, y)]
    off = 1.0 / H
    cols = []
    for p in range(3):
        u, v = uv[p]
        cols.append(_bilinear_zeros(planes[p, 0:3], u, v))
        cols.append(_bilinear_zeros(planes[p, 3:6], u + off, v))
        cols.append(_bilinear_zeros(planes[p, 6:9], u, v + off))
    return torch.cat(cols, dim=1)


def view_encoding(dirs):
    """dirs (M,3) -> (M,27): [d, sin(f d + ph)] for f in (1,1,2,2,4,4,8,8), ph in (0,pi/2)*4."""
    freqs = torch.repeat_interleave(2.0 ** torch.arange(4, dtype=torch.float32), 2)
    phases = torch.zeros(8)
    phases[1::2] = math.pi * 0.5
    e = torch.sin(phases[None, :, None] + dirs[:, None, :] * freqs[None, :, None])
    return torch.cat([dirs, e.reshape(dirs.shape[0], 24)], dim=1)


def mlp(p, feats, dirs=None):
    """Density (and colour) MLP. Returns sigma_raw (M,) or (rgb_raw (M,3), sigma_raw (M,))."""
    h = F.softplus(F.linear(feats, p["pts_linears.0.weight"], p["pts_linears.0.bias"]))
    h = F.softplus(F.linear(h, p["pts_linears.1.weight"], p["pts_linears.1.bias"]))
    h = torch.cat([feats, h], dim=1)  # skip after layer index 1, input first
    h = F.softplus(F.linear(h, p["pts_linears.2.weight"], p["pts_linears.2.bias"]))
    sigma = F.linear(h, p["alpha_linear.weight"], p["alpha_linear.bias"])[:, 0]
    if dirs is None:
        return sigma
    feat = F.linear(h, p["feature_linear.weight"], p["feature_linear.bias"])
    h = torch.cat([feat, view_encoding(dirs)], dim=1)
    h = F.softplus(F.linear(h, p["views_linear.weight"], p["views_linear.bias"]))
    rgb = F.linear(h, p["rgb_linear.weight"], p["rgb_linear.bias"])
    return rgb, sigma


def importance_z(sigma_raw, z, rays_d, u):
    """sigma_raw, z (R,N); u (R,Ni) -> sorted (R, N+Ni) merged depths."""
    R, N = z.shape
    dist = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), 1e10)], dim=1)
    dist = dist * rays_d.norm(dim=1, keepdim=True)
    alpha = 1.0 - torch.exp(-F.softplus(sigma_raw) * dist)
    trans = torch.cumprod(torch.cat([torch.ones(R, 1), 1.0 - alpha + 1e-10], dim=1), dim=1)[:, :-1]
    w = (alpha * trans)[:, 1:-1] + 1e-5
    pdf = w / w.sum(dim=1, keepdim=True)
    cdf = torch.cat([torch.zeros(R, 1), torch.cumsum(pdf, dim=1)], dim=1)  # (R, N-1)
    bins = 0.5 * (z[:, 1:] + z[:, :-1])  # (R, N-1)
    idx = torch.searchsorted(cdf, u.contiguous(), right=True)
    lo = (idx - 1).clamp(min=0)
    hi = idx.clamp(max=cdf.shape[1] - 1)
    c0, c1 = cdf.gather(1, lo), cdf.gather(1, hi)
    b0, b1 = bins.gather(1, lo), bins.gather(1, hi)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    znew = b0 + (u - c0) / den * (b1 - b0)
    return torch.sort(torch.cat([z, znew], dim=1), dim=1)[0]


def composite(rgb_raw, sigma_raw, z, white_bkgd=False, noise=None):
    """rgb_raw (R,S,3), sigma_raw, z (R,S) -> rgb (R,3), acc (R,), depth (R,).

    noise (R,S): the training-mode perturbation of the raw densities (renderer.py:212, `alpha + randn_like(alpha)`).
    """
    R = z.shape[0]
    if noise is not None:
        sigma_raw = sigma_raw + noise
    dist = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), 1e10)], dim=1)  # not scaled by |d|
    alpha = 1.0 - torch.exp(-F.softplus(sigma_raw) * dist)
    trans = torch.cumprod(torch.cat([torch.ones(R, 1), 1.0 - alpha + 1e-7], dim=1), dim=1)[:, :-1]
    w = alpha * trans
    acc = w.sum(dim=1)
    rgb = (torch.sigmoid(rgb_raw) * w[:, :, None]).sum(dim=1)
    if white_bkgd:
        rgb = rgb + (1.0 - acc[:, None])
    depth = (w * z).sum(dim=1)
    return rgb, acc, depth


def coarse_sigma(p, planes, bounds, rays_o, rays_d, z):
    R, N = z.shape
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
    return mlp(p, plane_features(planes, pts.reshape(-1, 3), bounds)).reshape(R, N)


def render_rays(p, planes, bounds, rays_o, rays_d, near, far, n_samples, n_importance, u=None,
                white_bkgd=False, normalize_depth=True, z_vals=None, return_aux=False, noise=None, clamp_depth=True):
    """One subject: planes (3,9,H,W), bounds (2,3), rays (R,3), near/far (R,).

    Returns rgb (R,3), acc (R,), depth (R,) [+ aux dict with sigma_coarse, z_all].
    noise (R, n_samples+n_importance): training mode (test=False).  The importance depths are computed under no_grad like
    the reference (renderer.py:243-253), so autograd through this function gives the reference's gradients for `planes` and `p`.
    clamp_depth=False: the recon_NeRF twin, which normalises the depth but does not clamp it (recon_NeRF/lib/renderer.py:288 vs
    human_diffusion/NeRF/renderer.py:272-274).
    """
    R = rays_o.shape[0]
    if z_vals is None:
        t = torch.linspace(0.0, 1.0, steps=n_samples)
        z = near[:, None] * (1.0 - t) + far[:, None] * t
    else:
        z = z_vals
    aux = {}
    if n_importance > 0:
        with torch.no_grad():
            sig = coarse_sigma(p, planes, bounds, rays_o, rays_d, z)
            z = importance_z(sig, z, rays_d, u)
        aux["sigma_coarse"] = sig
    aux["z_all"] = z
    S = z.shape[1]
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
    vd = rays_d / rays_d.norm(dim=1, keepdim=True)
    dirs = vd[:, None, :].expand(R, S, 3).reshape(-1, 3)
    rgb_raw, sig_f = mlp(p, plane_features(planes, pts.reshape(-1, 3), bounds), dirs)
    rgb, acc, depth = composite(rgb_raw.reshape(R, S, 3), sig_f.reshape(R, S), z, white_bkgd, noise)
    if normalize_depth:
        depth = (depth - near) / (far - near + 1e-5)
        if clamp_depth:
            depth = depth.clamp(0, 1)
    if return_aux:
        aux["sigma_fine"] = sig_f.reshape(R, S)
        aux["rgb_raw"] = rgb_raw.reshape(R, S, 3)
        return rgb, acc, depth, aux
    return rgb, acc, depth


def render_rays_recon(p, module_planes, tp_input, rays_o, rays_d, z_vals, near, far, n_importance, u, white_bkgd=False, noise=None):
    """recon_NeRF/lib/renderer.py:244-295.  module_planes (num_instances, 4, 3, 9, H, W) is the module's Parameter; tp_input carries
    'instance_idx', 'cloth_layer_index' (bs,) and 'world_bounds' (bs,2,3); rays (bs,R,3), z_vals (bs,R,N), near / far (bs,R) are the
    ARGUMENTS used for the depth normalisation only; u (bs,R,Ni), noise (bs,R,N+Ni) or None.
    Returns rgb (bs,R,3), acc (bs,R), depth (bs,R) - depth normalised, not clamped."""
    tri = module_planes[tp_input["instance_idx"], tp_input["cloth_layer_index"]]      # :251 (advanced indexing: a new tensor)
    outs = []
    for b in range(tri.shape[0]):
        outs.append(render_rays(p, tri[b], tp_input["world_bounds"][b], rays_o[b], rays_d[b], near[b], far[b], z_vals.shape[2],
                                n_importance, u=u[b] if u is not None else None, white_bkgd=white_bkgd, z_vals=z_vals[b],
                                noise=noise[b] if noise is not None else None, clamp_depth=False))
    return tuple(torch.stack([o[k] for o in outs]) for k in range(3))
