"""Differentiable rendering for the tri-plane fitting loop.

What the reference trains with (recon_NeRF/run_nerf_batch.py:236-265): Renderer.render in training mode (test=False; Gaussian noise on
the raw densities, renderer.py:212) -> MSE on rgb_map / acc_map -> loss.backward() -> Adam on the MLP and the tri-planes.  Autograd of
stock ops is replaced by one autograd.Function per subject whose forward and backward are the HIP kernels of hl_render.hip:

    forward    hl_render_eval_acts (coarse depths) -> hl_render_importance_new -> hl_render_eval_acts (new depths)
               -> hl_render_composite_noise                                    [saves the activation matrix]
    backward   hl_render_composite_backward -> hl_render_mlp_backward x2       [layer deltas]
               hl_render_plane_grads                                           [tri-plane gradient: transposed bilinear lookup]
               hl_render_weight_grads: weight gradient of every layer = deltas x activations^T over the sample points

Gradients exist for rgb_map (= normal_map, the same tensor as in the reference) and acc_map with respect to tri_planes and the seven
Linear layers.  depth_map is returned without gradient (the reference's losses never use it).  As in the reference the importance
depths are constants of the backward pass (torch.no_grad, renderer.py:243-253), and rays / depths / bounds get no gradient.

Canonical space (use_canonical_space=True, the TightCap fitting runs of README.md:123): geo["deform"] carries the per-subject tables of
NeRF/deform.py; every evaluate pass is preceded by hl_deform_rays and reads the canonical points (hl_render_eval_points_acts), and the
tri-plane gradient is scattered from those points (hl_render_plane_grads_points).  The deformation has no parameters.
"""
import ctypes as C

import torch

from .. import _lib


def _row_pad():
    """Extra floats per matrix row.  The number of sample points is a power of two in the usual configurations (2048 rays x 256 samples
    -> rows exactly 2 MiB apart), which sends the same column of every row to the same HBM channel; an odd multiple of 256 bytes spreads
    them (k_wgrad 1.05 -> 0.92 ms)."""
    return 96


def train_rows():
    a, d = C.c_int(0), C.c_int(0)
    _lib.lib().hl_render_train_rows(C.byref(a), C.byref(d))
    return a.value, d.value


class RenderRaysFunction(torch.autograd.Function):
    """One subject: planes (3,9,H,W) + the 14 MLP tensors (order of renderer._MLP_ORDER, weight then bias) -> rgb, acc, depth."""

    @staticmethod
    def forward(ctx, renderer, geo, planes, *mlp):
        L = _lib.lib()
        dev = planes.device
        ro, rd, nr, fr, bd, zb, ub, noise, flags = (geo[k] for k in ("rays_o", "rays_d", "near", "far", "bounds", "z", "u", "noise", "flags"))
        R, N = zb.shape
        Ni = ub.shape[1]
        H, W = planes.shape[-2:]
        T32 = (R + 31) // 32 * 32
        P = T32 * (N + Ni)
        act_rows, _ = train_rows()
        packed = renderer._packed_mlp(dev)
        pp = renderer._packed_planes(planes)
        LD = P + _row_pad()       # row pitch: see _row_pad
        act = torch.empty((act_rows, LD), dtype=torch.float32, device=dev)
        vc = torch.empty(T32 * N * 4, dtype=torch.float32, device=dev)
        vn = torch.empty(T32 * Ni * 4, dtype=torch.float32, device=dev)
        zn = torch.empty(T32 * Ni, dtype=torch.float32, device=dev)
        rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        acc = torch.empty((R,), dtype=torch.float32, device=dev)
        depth = torch.empty((R,), dtype=torch.float32, device=dev)
        p, st = _lib.ptr, _lib.stream_ptr()
        dfm = geo.get("deform")      # canonical space: (verts4, table, R host, Th host); `bounds` is then t_world_bounds
        pts = None
        if dfm is None:
            _lib.check(L.hl_render_eval_acts(p(packed), p(pp), H, W, p(bd), p(ro), p(rd), p(nr), p(fr), p(zb), 0, R, N, p(vc), p(act), LD, 0,
                                             st), "hl_render_eval_acts")
            _lib.check(L.hl_render_importance_new(p(vc), p(rd), p(nr), p(fr), p(zb), p(ub), R, N, Ni, p(zn), st), "hl_render_importance_new")
            _lib.check(L.hl_render_eval_acts(p(packed), p(pp), H, W, p(bd), p(ro), p(rd), p(nr), p(fr), p(zn), 1, R, Ni, p(vn), p(act), LD,
                                             T32 * N, st), "hl_render_eval_acts")
        else:
            verts4, table, Rh, Th = dfm
            nv = int(verts4.shape[0])
            pts = [torch.empty((T32 * n, 4), dtype=torch.float32, device=dev) for n in (N, Ni)]
            dirs = [torch.empty((T32 * n, 4), dtype=torch.float32, device=dev) for n in (N, Ni)]
            scr = torch.empty(4, dtype=torch.float32, device=dev)
            _lib.check(L.hl_deform_rays(p(ro), p(rd), p(nr), p(fr), p(zb), 0, R, N, Rh.ctypes.data, Th.ctypes.data, p(verts4), p(table), nv,
                                        p(pts[0]), p(dirs[0]), p(scr), st), "hl_deform_rays")
            _lib.check(L.hl_render_eval_points_acts(p(packed), p(pp), H, W, p(bd), p(pts[0]), p(dirs[0]), R, N, p(vc), p(act), LD, 0, st),
                       "hl_render_eval_points_acts")
            _lib.check(L.hl_render_importance_new(p(vc), p(rd), p(nr), p(fr), p(zb), p(ub), R, N, Ni, p(zn), st), "hl_render_importance_new")
            _lib.check(L.hl_deform_rays(p(ro), p(rd), p(nr), p(fr), p(zn), 1, R, Ni, Rh.ctypes.data, Th.ctypes.data, p(verts4), p(table), nv,
                                        p(pts[1]), p(dirs[1]), p(scr), st), "hl_deform_rays")
            _lib.check(L.hl_render_eval_points_acts(p(packed), p(pp), H, W, p(bd), p(pts[1]), p(dirs[1]), R, Ni, p(vn), p(act), LD, T32 * N, st),
                       "hl_render_eval_points_acts")
        ctx.pts = pts
        _lib.check(L.hl_render_composite_noise(p(nr), p(fr), p(zb), p(zn), p(vc), p(vn), p(noise), R, N, Ni, flags, p(rgb), p(acc),
                                               p(depth), st), "hl_render_composite_noise")
        ctx.renderer, ctx.geo, ctx.packed = renderer, geo, packed
        ctx.mlp_versions = [t._version for t in mlp]
        ctx.save_for_backward(act, vc, vn, zn, *mlp)
        ctx.hw = (H, W)
        ctx.mark_non_differentiable(depth)
        return rgb, acc, depth

    @staticmethod
    def backward(ctx, g_rgb, g_acc, _g_depth):
        L = _lib.lib()
        act, vc, vn, zn, *mlp = ctx.saved_tensors
        geo = ctx.geo
        ro, rd, nr, fr, bd, zb, ub, noise, flags = (geo[k] for k in ("rays_o", "rays_d", "near", "far", "bounds", "z", "u", "noise", "flags"))
        dev = act.device
        R, N = zb.shape
        Ni = ub.shape[1]
        H, W = ctx.hw
        T32 = (R + 31) // 32 * 32
        P = T32 * (N + Ni)
        _, del_rows = train_rows()
        g_rgb = (torch.zeros((R, 3), device=dev) if g_rgb is None else g_rgb).to(torch.float32).contiguous()
        g_acc = (torch.zeros((R,), device=dev) if g_acc is None else g_acc).to(torch.float32).contiguous()
        p, st = _lib.ptr, _lib.stream_ptr()
        d_rec = torch.empty((P, 4), dtype=torch.float32, device=dev)          # rows [0, T32*N): coarse pass, then the new depths
        dvc, dvn = d_rec[:T32 * N], d_rec[T32 * N:]
        LD = P + _row_pad()
        delta = torch.empty((del_rows, LD), dtype=torch.float32, device=dev)
        scratch = torch.empty(L.hl_render_composite_backward_scratch_bytes(R, N, Ni) // 4, dtype=torch.float32, device=dev)
        _lib.check(L.hl_render_composite_backward(p(nr), p(fr), p(zb), p(zn), p(vc), p(vn), p(noise), p(g_rgb), p(g_acc), R, N, Ni, flags,
                                                  p(dvc), p(dvn), p(delta), LD, p(scratch), st), "hl_render_composite_backward")
        # transposed weights of the values the forward pass used
        params = _lib.RenderMlpParams(*[C.c_void_p(t.data_ptr()) for t in mlp])
        bwd = torch.empty(L.hl_render_mlp_bwd_packed_bytes() // 4, dtype=torch.float32, device=dev)
        _lib.check(L.hl_render_mlp_pack_bwd(C.byref(params), p(bwd), st), "hl_render_mlp_pack_bwd")
        _lib.check(L.hl_render_mlp_backward(p(ctx.packed), p(bwd), H, W, p(bd), p(ro), p(rd), p(nr), p(fr), p(zb), 0, R, N, p(dvc), p(act), LD,
                                            0, p(delta), LD, 0, st), "hl_render_mlp_backward")
        _lib.check(L.hl_render_mlp_backward(p(ctx.packed), p(bwd), H, W, p(bd), p(ro), p(rd), p(nr), p(fr), p(zn), 1, R, Ni, p(dvn), p(act), LD,
                                            T32 * N, p(delta), LD, T32 * N, st), "hl_render_mlp_backward")
        d_planes = torch.empty((27, H, W), dtype=torch.float32, device=dev)
        if ctx.needs_input_grad[2] and ctx.pts is not None:
            sb = torch.empty(L.hl_render_plane_grads_points_scratch_bytes(R, N, Ni) // 4, dtype=torch.float32, device=dev)
            _lib.check(L.hl_render_plane_grads_points(H, W, p(bd), p(ctx.pts[0]), p(ctx.pts[1]), R, N, Ni, p(delta), LD, p(d_planes), p(sb), st),
                       "hl_render_plane_grads_points")
        elif ctx.needs_input_grad[2]:
            from .renderer import untile_rows
            zr = untile_rows(zn, R, Ni).contiguous()      # the scatter walks one ray per wave: give it the depths of a ray in one line
            sb = torch.empty(L.hl_render_plane_grads_scratch_bytes(R) // 4, dtype=torch.float32, device=dev)
            _lib.check(L.hl_render_plane_grads(H, W, p(bd), p(ro), p(rd), p(nr), p(fr), p(zb), p(zr), 1, R, N, Ni, p(delta), LD, p(d_planes),
                                               p(sb), st), "hl_render_plane_grads")
        # all 14 parameter gradients: rows of `delta` x rows of `act` over the sample points (include/humanliff_hip.h lists the rows)
        flat = torch.zeros(sum(t.numel() for t in mlp), dtype=torch.float32, device=dev)
        grads, o = [], 0
        for t in mlp:
            grads.append(flat[o:o + t.numel()].view(t.shape))
            o += t.numel()
        gp = _lib.RenderMlpParams(*[C.c_void_p(g.data_ptr()) for g in grads])
        wsb = torch.empty(L.hl_render_weight_grads_scratch_bytes(P) // 4, dtype=torch.float32, device=dev)    # the point ranges' partial results
        _lib.check(L.hl_render_weight_grads(p(delta), LD, p(act), LD, P, C.byref(gp), p(wsb), st), "hl_render_weight_grads")
        needs = ctx.needs_input_grad   # (renderer, geo, planes, *mlp)
        out = [None, None, d_planes.view(3, 9, H, W) if needs[2] else None]
        out += [g if needs[3 + i] else None for i, g in enumerate(grads)]
        return tuple(out)
