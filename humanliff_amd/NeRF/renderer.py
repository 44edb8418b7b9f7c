"""Drop-in `Renderer` / `render` for the tri-plane NeRF decode path, backed by the HIP kernels.

Mirrors (same names, argument order and return structure):
    Renderer.__init__   human_diffusion/NeRF/renderer.py:14-50
    Renderer.render     human_diffusion/NeRF/renderer.py:234-281   (recon twin: recon_NeRF/lib/renderer.py:244)
    render              human_diffusion/scripts/triplane_sample_layered.py:250-288
                        (recon twin: recon_NeRF/run_nerf_batch.py:29-67; "render_rays" in BASELINE.json)

Scope: triplane_ch=27; test=True (inference, the shipped SynBody sampling configuration) in world or canonical space, and
test=False (training mode with gradients, world space).  Everything numerical happens in libhumanliff_hip.so; there is no
PyTorch fallback (a missing library or a CPU tensor raises).
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from .fields import PositionalEncoding

def untile_rows(buf, n_rays, n_cols):
    """Tile-major workspace array [ceil(R/32)][n_cols][32] (include/humanliff_hip.h) -> rows (R, n_cols)."""
    tiles = (n_rays + 31) // 32
    return buf[:tiles * n_cols * 32].view(tiles, n_cols, 32).permute(0, 2, 1).reshape(tiles * 32, n_cols)[:n_rays]


def tile_rows(rows):
    """rows (R, n_cols) -> tile-major flat buffer, rays padded to a multiple of 32 with copies of the last row."""
    R, n_cols = rows.shape
    tiles = (R + 31) // 32
    pad = tiles * 32 - R
    if pad:
        rows = torch.cat([rows, rows[-1:].expand(pad, n_cols)], 0)
    return rows.view(tiles, 32, n_cols).permute(0, 2, 1).contiguous().view(-1)


_MLP_ORDER = ("pts_linears.0", "pts_linears.1", "pts_linears.2", "feature_linear", "alpha_linear",
              "views_linear", "rgb_linear")


_SIDE_STREAMS = {}   # device -> HIP streams of Renderer.subject_streams (process-wide pool)


class Renderer(nn.Module):
    def __init__(self, use_canonical_space=True, num_instances=1, triplane_dim=256, triplane_ch=18,
                 smpl_type='smpl', test=False):
        super().__init__()
        self.use_canonical_space = use_canonical_space
        self.num_instances = num_instances
        self.triplane_dim = triplane_dim
        self.triplane_ch = triplane_ch
        self.test = test
        self.smpl_type = smpl_type
        self.view_enc = PositionalEncoding(num_freqs=4)
        d_in, d_hidden, n_layers = triplane_ch, 128, 2
        self.skips = [n_layers / 2]
        widths = [d_in] + [d_hidden + d_in if i in self.skips else d_hidden for i in range(n_layers)]
        self.pts_linears = nn.ModuleList([nn.Linear(w, d_hidden) for w in widths])
        self.feature_linear = nn.Linear(d_hidden, d_hidden)
        self.alpha_linear = nn.Linear(d_hidden, 1)
        self.views_linear = nn.Linear(d_hidden + 27, d_hidden // 2)
        self.rgb_linear = nn.Linear(d_hidden // 2, 3)
        # The reference loads the SMPL(-X) asset here (renderer.py:41-50: assets/SMPL_NEUTRAL.pkl -> SMPL_to_tensor); it is only
        # used by the canonical-space deformation.  The asset is licensed data and not shipped: assign the dict of tensors
        # (keys v_template, shapedirs, posedirs, J_regressor, kintree_table, weights) to `SMPL_NEUTRAL` before rendering with
        # use_canonical_space=True.
        self.SMPL_NEUTRAL = None
        # depth_map = (depth - near) / (far - near + 1e-5), clamped to [0,1] by this twin (renderer.py:271-274); the recon_NeRF twin
        # (recon_NeRF/lib/renderer.py:288) does not clamp and clears the second flag
        self._depth_flags = _lib.HL_RENDER_NORMALIZE_DEPTH | _lib.HL_RENDER_CLAMP_DEPTH
        self.uniforms_on_device = False      # extension: draw sample_pdf's uniforms with the device generator (see render())
        self.cpu_uniforms_on_host = False    # True: draw the reference's CPU-generator uniforms on the host and upload them (renderer.py:545 literally);
                                             # default: the SAME numbers continued on the device from the CPU generator's state (NeRF/cpu_rng.py)
        self.subject_streams = True          # training mode puts subjects after the first on their own HIP streams (see _render_training): one subject's HBM-bound
                                             # kernels overlap the other's matrix-bound ones, +9 % on the fitting step.  Since round 5 the backward has no float
                                             # atomics, so images and gradients are the same BITS with the switch on or off (tests/test_render_train_gpu.py); PyTorch warns
                                             # once that the parameters' AccumulateGrad nodes sit on another stream than the incoming gradients.  False: one stream.
        self.mlp_fp16 = False                # extension, opt-in: the MLP with fp16 operands / fp32 accumulation (HL_RENDER_MLP_FP16, k_march16) in
                                             # render() without canonical space; canonical-space rendering, density_grid() and training stay fp32
        self.mlp_products = "fp16x2"         # how render() forms the fp32 products of the MLP in the evaluate-once pipeline (test mode, world space):
                                             # "fp16x2" (default, round 5) = both operands as TWO fp16 planes (activations h0 + h1 to 2^-20, weights
                                             # nearest-even to 2^-22), the three partial products h0 w0 + h0 w1 + h1 w0 on v_mfma_f32_32x32x16_f16, fp32
                                             # accumulation (k_march_plw<2>, HL_RENDER_MLP_FP16X2); "bf16x3" = both operands split EXACTLY into three bf16
                                             # planes, six partial products (k_march_plw<3>, HL_RENDER_MLP_BF16X3); "fp32" = v_mfma_f32_32x32x2_f32
                                             # (k_march).  Same tolerances, tested on all three; rgb of the three within 7e-7 of each other on full views.
        self._ws = None

    # ---- packing caches ------------------------------------------------------------------------
    def _mlp_tensors(self):
        sd = dict(self.named_parameters())
        out = []
        for stem in _MLP_ORDER:
            out += [sd[stem + ".weight"], sd[stem + ".bias"]]
        return out

    def _packed_mlp(self, device):
        """The MLP re-laid for the kernels (17 chunks in MFMA A-fragment order), rebuilt on EVERY call: a 5 us kernel.  An
        (address, version) keyed cache served stale weights to a fitting loop whose optimizer is `Adam(..., fused=True)` - the fused
        optimizers (and any writer that goes through raw pointers) update parameters without touching `Tensor._version`, so the
        loss simply stopped moving (scripts/adam_check.py; tests/test_render_train_gpu.py pins it)."""
        ts = self._mlp_tensors()
        for t in ts:
            if not t.is_cuda:
                raise RuntimeError("Renderer parameters must live on the GPU (call .to('cuda')); "
                                   "humanliff_amd has no CPU path")
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("Renderer parameters must be contiguous fp32")
        L = _lib.lib()
        params = _lib.RenderMlpParams(*[C.c_void_p(t.data_ptr()) for t in ts])
        buf = torch.empty(L.hl_render_mlp_packed_bytes() // 4, dtype=torch.float32, device=device)
        _lib.check(L.hl_render_mlp_pack(C.byref(params), _lib.ptr(buf), _lib.stream_ptr()), "hl_render_mlp_pack")
        return buf

    def _packed_planes(self, planes):
        """planes (3,9,H,W) fp32 device view of one subject -> packed texel-major copy.

        Repacked on every call (17 MB of traffic, a few microseconds next to a 75 ms view or an 8 ms fitting step).  An address /
        version keyed cache is unsafe here: callers hand in freshly allocated tensors (`self.tri_planes[idx, layer]` of the recon
        twin, `sample.clamp().reshape()` of the sampling script) whose `_version` is 0 and whose storage address the caching
        allocator reuses as soon as the previous one is freed - different contents would hit a stale packed copy."""
        if not planes.is_cuda:
            raise RuntimeError("tri_planes must live on the GPU; humanliff_amd has no CPU path")
        if planes.dtype != torch.float32:
            raise RuntimeError(f"tri_planes must be float32 (got {planes.dtype})")
        L = _lib.lib()
        H, W = planes.shape[-2:]
        src = planes.detach().contiguous()
        buf = torch.empty(L.hl_planes_packed_bytes(H, W) // 4, dtype=torch.float32, device=planes.device)
        _lib.check(L.hl_planes_pack(_lib.ptr(src), H, W, _lib.ptr(buf), _lib.stream_ptr(planes.device)), "hl_planes_pack")
        return buf

    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() * 4 < nbytes or self._ws.device != device:
            self._ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        return self._ws

    # ---- reference API ---------------------------------------------------------------------------
    def render(self, tp_input, world_pts, z_vals, rays_o, rays_d, near, far, tri_planes, n_importance=128,
               white_bkgd=False, *, n_samples=None, u=None, reevaluate=False, noise=None):
        """Same contract as the reference's Renderer.render; returns the dict
        {'rgb_map','acc_map','normal_map','depth_map'} of detached tensors.

        world_pts is accepted for signature compatibility and not read: the kernel forms
        rays_o + rays_d * z_vals itself, which is how every reference caller builds world_pts
        (triplane_sample_layered.py:277).  Extensions (keyword-only): z_vals=None with n_samples
        lets the kernel generate the linspace depths; u (bs*R, n_importance) supplies sample_pdf's
        uniforms, otherwise they are drawn exactly like the reference does - torch.rand on the
        CPU generator, then copied to the device (renderer.py:545).  reevaluate=True makes the fine pass run
        the network on all n_samples + n_importance depths like the reference (re-evaluating the coarse
        points); by default every point is evaluated once and the two sorted halves are merged - 23 % less arithmetic.  With
        mlp_products="fp32" the two schedules give bit-identical images; the split products ("fp16x2", the default, and "bf16x3") exist in the evaluate-once
        kernel only, so reevaluate=True runs the fp32-MFMA kernel (as do canonical-space rendering, density_grid() and training) and its
        images differ from the default's within the fp32 tolerance (rgb 7e-7).  Precedence: mlp_fp16=True (opt-in, NOT fp32 tolerance)
        overrides mlp_products.

        test=False (training mode, renderer.py:212, 280): the fine pass adds Gaussian noise to the raw densities (`noise`
        (bs*R*(n_samples+n_importance), 1) supplies it, otherwise it is drawn on the device like the reference's randn_like) and
        rgb_map / acc_map stay attached to the autograd graph of tri_planes and the MLP parameters - see NeRF/train.py.
        """
        if not tri_planes.is_cuda:
            raise RuntimeError("Renderer.render needs CUDA(HIP) tensors; there is no CPU path")
        with _lib.on(tri_planes.device):      # launches go to the tri-planes' device, whatever the caller's current device is
            return self._render(tp_input, z_vals, rays_o, rays_d, near, far, tri_planes, n_importance, white_bkgd, n_samples, u,
                                reevaluate, noise)

    def _render(self, tp_input, z_vals, rays_o, rays_d, near, far, tri_planes, n_importance, white_bkgd, n_samples, u, reevaluate, noise):
        if self.use_canonical_space and self.SMPL_NEUTRAL is None:
            raise RuntimeError("use_canonical_space=True needs the body model: set renderer.SMPL_NEUTRAL to the SMPL_to_tensor dict")
        if self.use_canonical_space and self.test:
            return self._render_canonical(tp_input, z_vals, rays_o, rays_d, near, far, tri_planes, n_importance, white_bkgd,
                                          n_samples, u)
        if self.triplane_ch != 27:
            raise NotImplementedError("the HIP ray-march kernel is specialised for triplane_ch=27")
        assert tri_planes.dim() == 5 and tri_planes.shape[1] == 3 and tri_planes.shape[2] == 9, \
            "tri_planes must be (bs, 3, 9, H, W)"
        bs, _, _, H, W = tri_planes.shape
        dev = tri_planes.device
        rays_o = rays_o.reshape(bs, -1, 3)
        rays_d = rays_d.reshape(bs, -1, 3)
        R = rays_o.shape[1]
        near = near.reshape(bs, R)
        far = far.reshape(bs, R)
        if z_vals is not None:
            assert z_vals.shape[:2] == (bs, R)
            n_samples = z_vals.shape[2]
        assert n_samples is not None and n_samples >= 2
        bounds = tp_input['t_world_bounds' if self.use_canonical_space else 'world_bounds'].reshape(bs, 2, 3)
        pending_draw = None
        if n_importance > 0:
            assert n_importance == n_samples, \
                "the reference reshapes coarse densities to n_importance (renderer.py:250): counts must match"
            if u is None:
                # the reference draws sample_pdf's uniforms on the CPU generator and uploads them (renderer.py:545); `uniforms_on_device`
                # (an extension, off by default) draws them on the device instead: same distribution, no 2 MB upload per fitting step
                # The CPU draw lands in pinned memory and is uploaded asynchronously: the same values from the same generator, without
                # the stream-draining synchronous copy of pageable memory that cost the fitting loop its run-ahead (0.8 ms per step).
                if self.uniforms_on_device:
                    u = torch.rand([bs * R, n_importance], device=dev)
                elif not getattr(self, "cpu_uniforms_on_host", False) and bs * R * n_importance >= (1 << 16):
                    # the reference's numbers (torch.rand of the CPU generator), written by the device: the host generator is advanced below,
                    # once everything of this call is enqueued (the draw of a 512x512 view costs the host 50 - 80 ms, the device runs it next to
                    # the coarse pass; a fitting step's 0.5 M numbers cost the host ~1 ms and a 2 MB upload).  Test mode: only the importance
                    # launch waits for the generator (hl_render_rays_u_event); training mode: the current stream waits (the generator runs on its
                    # own stream, beside the tail of the previous step).
                    from .cpu_rng import rand_like_cpu
                    u, pending_draw = rand_like_cpu([bs * R, n_importance], dev, defer_wait=self.test)
                else:
                    u = torch.rand([bs * R, n_importance], pin_memory=True).to(dev, non_blocking=True)
            u = u.reshape(bs, R, n_importance)
        # Everything behind the draw runs inside try / finally: whatever happens (a launch that raises included), the CPU generator ends up where the
        # reference's torch.rand would have left it - a failed call must not make the next one reuse the same uniforms.
        try:
            if not self.test:
                out = self._render_training(tri_planes, bounds, z_vals, rays_o, rays_d, near, far, n_samples, n_importance, white_bkgd, u,
                                            noise, tp_input if self.use_canonical_space else None)
                return out
            L = _lib.lib()
            packed = self._packed_mlp(dev)
            ws = self._workspace(L.hl_render_workspace_bytes(R, n_samples, n_importance), dev)
            rgb = torch.empty((bs, R, 3), dtype=torch.float32, device=dev)
            acc = torch.empty((bs, R), dtype=torch.float32, device=dev)
            depth = torch.empty((bs, R), dtype=torch.float32, device=dev)
            products = getattr(self, "mlp_products", "fp32")
            if products not in ("fp16x2", "bf16x3", "fp32"):
                raise ValueError(f"Renderer.mlp_products must be 'fp16x2', 'bf16x3' or 'fp32', not {products!r}")
            fp16 = bool(getattr(self, "mlp_fp16", False))
            # explicit precedence: the opt-in fp16-operand kernel first; the bf16x3 products only in the evaluate-once schedule (the re-evaluating
            # one exists on the fp32-MFMA kernel alone)
            # (Renderer.four_launch, a developer / test switch: rounds 2-5's evaluate / k_importance / evaluate / k_composite schedule instead of the two-launch one-pass form)
            flags = self._depth_flags | (_lib.HL_RENDER_WHITE_BKGD if white_bkgd else 0) | (_lib.HL_RENDER_REEVALUATE if reevaluate else 0) | \
                (_lib.HL_RENDER_FOUR_LAUNCH if getattr(self, "four_launch", False) else 0) | \
                (_lib.HL_RENDER_MLP_FP16 if fp16 else (0 if reevaluate else {"bf16x3": _lib.HL_RENDER_MLP_BF16X3, "fp16x2": _lib.HL_RENDER_MLP_FP16X2}.get(products, 0)))
            f32 = lambda t: t.to(torch.float32).contiguous()  # noqa: E731
            for b in range(bs):
                pp = self._packed_planes(tri_planes[b])
                zb = f32(z_vals[b]) if z_vals is not None else None
                ub = f32(u[b]) if n_importance > 0 else None
                ro, rd, nr, fr, bd = f32(rays_o[b]), f32(rays_d[b]), f32(near[b]), f32(far[b]), f32(bounds[b])
                # (u drawn by the device on a side stream: only the importance-sampling launch waits for it, the coarse pass runs beside the generator)
                u_ev = pending_draw.u_event.cuda_event if pending_draw is not None else None
                _lib.check(L.hl_render_rays_u_event(
                    _lib.ptr(packed), _lib.ptr(pp), H, W, _lib.ptr(bd), _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(nr),
                    _lib.ptr(fr), _lib.ptr(zb), _lib.ptr(ub), u_ev, R, n_samples, n_importance, flags,
                    _lib.ptr(rgb[b]), _lib.ptr(acc[b]), _lib.ptr(depth[b]), _lib.ptr(ws), _lib.stream_ptr()),
                    "hl_render_rays")
            # normal_map aliases rgb_map in the reference (renderer.py:228)
            return {'rgb_map': rgb, 'acc_map': acc, 'normal_map': rgb, 'depth_map': depth}
        finally:
            if pending_draw is not None:
                pending_draw.finish()


    # ---- training mode (test=False) ------------------------------------------------
    @staticmethod
    def _train_ray_chunk(samples_per_ray):
        return max(32, ((2 ** 31 - 1) // (4 * 630 * samples_per_ray)) // 32 * 32)

    def _render_training(self, tri_planes, bounds, z_vals, rays_o, rays_d, near, far, n_samples, n_importance, white_bkgd, u,
                         noise=None, tp_canonical=None):
        """Renderer.render with test=False (renderer.py:212, 276-281): Gaussian noise on the raw densities of the fine pass and
        outputs attached to the autograd graph - gradients for tri_planes and the MLP come from the HIP backward kernels
        (NeRF/train.py).  The noise is drawn like the reference's randn_like: one (bs*R*(n_samples+n_importance), 1) draw on the device."""
        assert n_importance > 0, "training mode is built for the hierarchical schedule (n_importance = n_samples)"
        bs, R = rays_o.shape[:2]
        dev = tri_planes.device
        S = n_samples + n_importance
        if z_vals is None:
            t = torch.linspace(0., 1., steps=n_samples, device=dev)
            z_vals = near[..., None] * (1. - t) + far[..., None] * t
        if noise is None:
            noise = torch.randn((bs * R * S, 1), device=dev)
        noise = noise.reshape(bs, R, S)
        flags = self._depth_flags | (_lib.HL_RENDER_WHITE_BKGD if white_bkgd else 0)
        mlp = self._mlp_tensors()
        # the activation / delta matrices of one call are addressed with 32-bit offsets and kept below 2 GiB (630 rows x 4 B per
        # sample point): 2048 rays x 256 samples need 1.3 GB; larger ray batches go down in pieces (rays are independent)
        rc = self._train_ray_chunk(S)
        outs = []
        # Subjects are independent until the loss: subject b > 0 goes to its own HIP stream (forward here, and - because autograd runs a
        # node's backward on the stream of its forward - backward too), so one subject's HBM-bound kernels (k_wgrad, k_plane_scatter)
        # overlap the other's matrix-bound ones (k_march, k_mlp_bwd).  Tensors of the caller's stream that a side stream reads are
        # recorded on it (the caching allocator must not hand their memory out while the side stream still reads).
        main = torch.cuda.current_stream(dev)
        side = self._subject_streams(dev, bs - 1) if getattr(self, "subject_streams", False) else []
        for st in side:                   # fork before anything of this call is queued: a side stream waits for the inputs only
            st.wait_stream(main)
            tri_planes.record_stream(st)
        for b in range(bs):
            st = side[b - 1] if (b > 0 and side) else None
            if st is None:
                outs.append(self._train_one_subject(b, None, tri_planes, bounds, z_vals, rays_o, rays_d, near, far, u, noise, flags, mlp, rc,
                                                    tp_canonical))
                continue
            with torch.cuda.stream(st):
                outs.append(self._train_one_subject(b, st, tri_planes, bounds, z_vals, rays_o, rays_d, near, far, u, noise, flags, mlp, rc,
                                                    tp_canonical))
        for b in range(1, bs if side else 0):
            main.wait_stream(side[b - 1])
            for t in outs[b]:
                t.record_stream(main)
        rgb = torch.stack([o[0] for o in outs])
        acc = torch.stack([o[1] for o in outs])
        depth = torch.stack([o[2] for o in outs])
        return {'rgb_map': rgb, 'acc_map': acc, 'normal_map': rgb, 'depth_map': depth}

    def _subject_streams(self, dev, n):
        # (the pool lives outside the module: stream objects in an nn.Module's __dict__ would ride along into copy.deepcopy / torch.save)
        have = _SIDE_STREAMS.setdefault(dev, [])
        while len(have) < n:
            have.append(torch.cuda.Stream(device=dev))
        return have[:n]

    def _train_one_subject(self, b, side, tri_planes, bounds, z_vals, rays_o, rays_d, near, far, u, noise, flags, mlp, rc, tp_canonical):
        from .train import RenderRaysFunction
        dev = tri_planes.device
        R = rays_o.shape[1]

        def f32(t):
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            if side is not None:
                t.record_stream(side)
            return t
        dfm = None
        if tp_canonical is not None:      # per-subject deformation tables (NeRF/deform.py); `bounds` is t_world_bounds here
            from .deform import deform_tables
            if self.SMPL_NEUTRAL is None:
                raise RuntimeError("use_canonical_space=True needs the body model: set renderer.SMPL_NEUTRAL to the SMPL_to_tensor dict")
            def rec(t):                   # inputs of the caller's stream that this subject's (side) stream reads
                if side is not None and torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(side)
                return t
            one = lambda d: {k: rec(v[b:b + 1]) for k, v in d.items()}  # noqa: E731
            for v in self.SMPL_NEUTRAL.values():
                rec(v)
            dfm = deform_tables(self.SMPL_NEUTRAL, one(tp_canonical['params']), one(tp_canonical['t_params']),
                                rec(tp_canonical['vertices'][b:b + 1].to(dev)))
        parts = []
        for i in range(0, R, rc):
            sl = slice(i, min(R, i + rc))
            geo = {"deform": dfm, "rays_o": f32(rays_o[b, sl]), "rays_d": f32(rays_d[b, sl]), "near": f32(near[b, sl]), "far": f32(far[b, sl]),
                   "bounds": f32(bounds[b]), "z": f32(z_vals[b, sl]), "u": f32(u[b, sl]), "noise": f32(noise[b, sl]), "flags": flags}
            parts.append(RenderRaysFunction.apply(self, geo, tri_planes[b], *mlp))
        return parts[0] if len(parts) == 1 else tuple(torch.cat([q[k] for q in parts]) for k in range(3))

    # ---- rendering through the canonical-space deformation -------------------------
    def _render_canonical(self, tp_input, z_vals, rays_o, rays_d, near, far, tri_planes, n_importance, white_bkgd, n_samples, u):
        """use_canonical_space=True (renderer.py:114-132, 192-201, 242-246): every sample point and its unit ray direction go through
        deform_target2c (1-NN body vertex, inverse LBS, blend-shape offsets, forward LBS into the big pose) before the tri-plane
        lookup in tp_input['t_world_bounds'].  One posed subject per call (batch 1)."""
        from .deform import deform_tables
        if self.SMPL_NEUTRAL is None:
            raise RuntimeError("use_canonical_space=True needs the body model: set renderer.SMPL_NEUTRAL to the SMPL_to_tensor dict")
        if self.triplane_ch != 27:
            raise NotImplementedError("the HIP ray-march kernel is specialised for triplane_ch=27")
        assert tri_planes.dim() == 5 and tri_planes.shape[:3] == (1, 3, 9), "tri_planes must be (1, 3, 9, H, W)"
        _, _, _, H, W = tri_planes.shape
        dev = tri_planes.device
        rays_o, rays_d = rays_o.reshape(1, -1, 3), rays_d.reshape(1, -1, 3)
        R = rays_o.shape[1]
        near, far = near.reshape(1, R), far.reshape(1, R)
        if z_vals is not None:
            n_samples = z_vals.shape[2]
        assert n_samples is not None and n_importance == n_samples, \
            "the reference reshapes coarse densities to n_importance (renderer.py:250): counts must match"
        pending_draw = None
        if u is None:
            if getattr(self, "cpu_uniforms_on_host", False) or R * n_importance < (1 << 16):
                u = torch.rand([R, n_importance]).to(dev)
            else:
                from .cpu_rng import rand_like_cpu
                u, pending_draw = rand_like_cpu([R, n_importance], dev)
        try:                                      # (the generator is advanced whatever happens below: see render)
            verts4, table, Rh, Th = deform_tables(self.SMPL_NEUTRAL, tp_input['params'], tp_input['t_params'],
                                                  tp_input['vertices'].to(dev))
            L = _lib.lib()
            packed, pp = self._packed_mlp(dev), self._packed_planes(tri_planes[0])
            ws = self._workspace(L.hl_render_canonical_workspace_bytes(R, n_samples, n_importance), dev)
            rgb = torch.empty((1, R, 3), dtype=torch.float32, device=dev)
            acc = torch.empty((1, R), dtype=torch.float32, device=dev)
            depth = torch.empty((1, R), dtype=torch.float32, device=dev)
            flags = self._depth_flags | (_lib.HL_RENDER_WHITE_BKGD if white_bkgd else 0)
            f32 = lambda t: t.to(torch.float32).contiguous()  # noqa: E731
            ro, rd, nr, fr = f32(rays_o[0]), f32(rays_d[0]), f32(near[0]), f32(far[0])
            bd = f32(tp_input['t_world_bounds'].reshape(-1, 2, 3)[0].to(dev))
            zb = f32(z_vals[0]) if z_vals is not None else None
            ub = f32(u.reshape(R, n_importance))
            _lib.check(L.hl_render_rays_canonical(
                _lib.ptr(packed), _lib.ptr(pp), H, W, _lib.ptr(bd), _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(nr), _lib.ptr(fr), _lib.ptr(zb),
                _lib.ptr(ub), R, n_samples, n_importance, flags, Rh.ctypes.data, Th.ctypes.data, _lib.ptr(verts4), _lib.ptr(table),
                int(verts4.shape[0]), _lib.ptr(rgb), _lib.ptr(acc), _lib.ptr(depth), _lib.ptr(ws), _lib.stream_ptr()),
                "hl_render_rays_canonical")
            return {'rgb_map': rgb, 'acc_map': acc, 'normal_map': rgb, 'depth_map': depth}
        finally:
            if pending_draw is not None:
                pending_draw.finish()

    # ---- the density grid behind extract_geometry ---------------------
    def density_grid(self, tp_input, tri_planes=None, resolution=512, rays_per_launch=1 << 17):
        """u[x,y,z] = -sigma_raw at the lattice linspace(bound_min, bound_max, resolution)^3, exactly the field
        the reference hands to marching cubes (renderer.py:290-321) - 134 M density-MLP evaluations at 512^3.

        Runs on the coarse ray-march kernel: lattice column (x, y) is a "ray" with origin (X[x], Y[y], 0),
        direction (0, 0, 1) and explicit depths Z, so the sample points are bit-identical to the reference's
        meshgrid.  Returns a (resolution,)*3 fp32 tensor on the tri-plane's device.
        """
        assert tri_planes is not None and tri_planes.shape[0] == 1 and tri_planes.shape[1:3] == (3, 9)
        with _lib.on(tri_planes.device):
            return self._density_grid(tp_input, tri_planes, resolution, rays_per_launch)

    def _density_grid(self, tp_input, tri_planes, resolution, rays_per_launch):
        dev = tri_planes.device
        H, W = tri_planes.shape[-2:]
        N = int(resolution)
        bounds = tp_input['world_bounds'].reshape(-1, 2, 3)[0].to(device=dev, dtype=torch.float32).contiguous()
        lo, hi = bounds[0].cpu(), bounds[1].cpu()
        X = torch.linspace(float(lo[0]), float(hi[0]), N)
        Y = torch.linspace(float(lo[1]), float(hi[1]), N)
        Z = torch.linspace(float(lo[2]), float(hi[2]), N)
        xx, yy = torch.meshgrid(X, Y, indexing="ij")
        rays_o = torch.stack([xx.reshape(-1), yy.reshape(-1), torch.zeros(N * N)], dim=1).to(dev)
        rays_d = torch.tensor([0.0, 0.0, 1.0]).expand(N * N, 3).contiguous().to(dev)
        zero = torch.zeros(N * N, device=dev)
        L = _lib.lib()
        packed, pp = self._packed_mlp(dev), self._packed_planes(tri_planes[0])
        out = torch.empty((N * N, N), dtype=torch.float32, device=dev)
        rays_per_launch = max(32, rays_per_launch // 32 * 32)
        T32 = ((rays_per_launch + 31) // 32) * 32
        if self.use_canonical_space:
            # renderer.py:311-314: the lattice spans world_bounds, every point goes through deform_target2c and is looked up in
            # t_world_bounds.  Same column-rays; hl_deform_rays writes their canonical points, hl_render_eval_points evaluates them
            # (full MLP: the density is the first value of the record)
            from .deform import deform_tables
            if self.SMPL_NEUTRAL is None:
                raise RuntimeError("use_canonical_space=True needs the body model: set renderer.SMPL_NEUTRAL to the SMPL_to_tensor dict")
            verts4, table, Rh, Th = deform_tables(self.SMPL_NEUTRAL, tp_input['params'], tp_input['t_params'], tp_input['vertices'].to(dev))
            tb = tp_input['t_world_bounds'].reshape(-1, 2, 3)[0].to(device=dev, dtype=torch.float32).contiguous()
            pc, dc = torch.empty((T32 * N, 4), device=dev), torch.empty((T32 * N, 4), device=dev)
            rec, scr = torch.empty((T32 * N, 4), device=dev), torch.empty(4, device=dev)
            for i in range(0, N * N, rays_per_launch):
                j = min(N * N, i + rays_per_launch)
                z = Z.to(dev)[None].expand(j - i, N).contiguous()
                ro, rd = rays_o[i:j].contiguous(), rays_d[i:j].contiguous()
                _lib.check(L.hl_deform_rays(_lib.ptr(ro), _lib.ptr(rd), _lib.ptr(zero[i:j]), _lib.ptr(zero[i:j]), _lib.ptr(z), 0, j - i, N,
                                            Rh.ctypes.data, Th.ctypes.data, _lib.ptr(verts4), _lib.ptr(table), int(verts4.shape[0]), _lib.ptr(pc),
                                            _lib.ptr(dc), _lib.ptr(scr), _lib.stream_ptr()), "hl_deform_rays")
                _lib.check(L.hl_render_eval_points(_lib.ptr(packed), _lib.ptr(pp), H, W, _lib.ptr(tb), _lib.ptr(pc), _lib.ptr(dc), j - i, N,
                                                   _lib.ptr(rec), _lib.stream_ptr()), "hl_render_eval_points")
                out[i:j] = untile_rows(rec[:, 0].contiguous(), j - i, N)
            return (-out).reshape(N, N, N)
        tmp = torch.empty(T32 * N, dtype=torch.float32, device=dev)
        # (131 072 columns per launch = 512 workgroups: with 32 768 - 128 workgroups - half of the CUs sat idle, 512^3 took 184 ms)
        zfull = Z.to(dev)[None].expand(min(rays_per_launch, N * N), N).contiguous()
        for i in range(0, N * N, rays_per_launch):
            j = min(N * N, i + rays_per_launch)
            z = zfull[:j - i]
            ro, rd = rays_o[i:j].contiguous(), rays_d[i:j].contiguous()
            _lib.check(L.hl_render_coarse(_lib.ptr(packed), _lib.ptr(pp), H, W, _lib.ptr(bounds), _lib.ptr(ro), _lib.ptr(rd),
                                          _lib.ptr(zero[i:j]), _lib.ptr(zero[i:j]), _lib.ptr(z), j - i, N, _lib.ptr(tmp),
                                          _lib.stream_ptr()), "hl_render_coarse")
            out[i:j] = untile_rows(tmp, j - i, N)
        return (-out).reshape(N, N, N)

    def extract_geometry(self, tp_input, tri_planes=None, resolution=512, threshold=0.0):
        """Reference signature (renderer.py:290).  The density field comes from the HIP kernel; smoothing and
        marching cubes stay on the CPU in PyMCubes, as in the reference (an external dependency, not rebuilt)."""
        try:
            import mcubes
        except ImportError as e:
            raise ImportError("extract_geometry needs PyMCubes for marching cubes (as the reference does); "
                              "Renderer.density_grid() provides the density field without it") from e
        u = self.density_grid(tp_input, tri_planes, resolution).cpu().numpy()
        vertices, triangles = mcubes.marching_cubes(mcubes.smooth(u), threshold)
        b = tp_input['world_bounds'].reshape(-1, 2, 3)[0].detach().cpu().numpy()
        vertices = vertices / (resolution - 1.0) * (b[1] - b[0])[None, :] + b[0][None, :]
        return vertices, triangles


def render(chunk=1024 * 32, rays_o=None, rays_d=None, near=0., far=1., tri_planes=None, tp_input=None, renderer=None,
           n_samples=128, perturb=0., n_importance=0, white_bkgd=False):
    """Render rays; returns [rgb_map, acc_map, normal_map, depth_map] like the reference.

    The reference walks the rays in `chunk`-sized pieces to bound its intermediates (~2.4 KB per sample
    point); the fused kernel keeps them on chip, so all rays go down in ONE launch set (16 384-ray chunks
    would fill only a quarter of the MI355X).  Rays are independent, so the result per ray is unchanged;
    the random draws keep the reference's per-chunk call pattern (same shapes, same order): sample_pdf's
    uniforms from the CPU generator (renderer.py:545) and, for perturb > 0, the stratified jitter from the
    device generator (triplane_sample_layered.py:275).
    """
    batch_size, n_rays, _ = rays_d.shape
    rays_o = rays_o.reshape(batch_size, -1, 3)
    rays_d = rays_d.reshape(batch_size, -1, 3)
    near = near.reshape(batch_size, -1, 1)
    far = far.reshape(batch_size, -1, 1)
    # the recon twin passes a DDP/DataParallel-wrapped renderer (run_nerf_batch.py:58)
    core = renderer.module if hasattr(renderer, "module") else renderer
    R = rays_o.shape[1]
    z, us = None, []
    if perturb > 0.:
        zs = []
        t_vals = torch.linspace(0., 1., steps=n_samples, device=rays_o.device)
        for i in range(0, R, chunk):
            zc = near[:, i:i + chunk] * (1. - t_vals) + far[:, i:i + chunk] * t_vals
            mids = .5 * (zc[..., 1:] + zc[..., :-1])
            upper = torch.cat([mids, zc[..., -1:]], -1)
            lower = torch.cat([zc[..., :1], mids], -1)
            zs.append(lower + (upper - lower) * torch.rand(zc.shape, device=rays_o.device))
        z = torch.cat(zs, 1)
    if n_importance > 0 and getattr(core, "uniforms_on_device", False):
        u = torch.rand([batch_size, R, n_importance], device=rays_o.device)      # extension, see Renderer.render
    elif n_importance > 0 and core.test and not getattr(core, "cpu_uniforms_on_host", False) and batch_size * R * n_importance >= (1 << 16) \
            and rays_o.is_cuda:
        # the reference's per-chunk draws (torch.rand([bs * chunk_rays, n_importance]) per chunk) are consecutive pieces of ONE serial stream:
        # the device continues the CPU generator's stream for all of them and the pieces are put where the chunks would have put them
        from .cpu_rng import rand_like_cpu
        flat, pending = rand_like_cpu([batch_size * R * n_importance], rays_o.device)
        pieces, o = [], 0
        for i in range(0, R, chunk):
            cr = min(chunk, R - i)
            pieces.append(flat[o:o + batch_size * cr * n_importance].reshape(batch_size, cr, n_importance))
            o += batch_size * cr * n_importance
        u = pieces[0] if len(pieces) == 1 else torch.cat(pieces, 1)
        pending.finish()
    elif n_importance > 0:
        for i in range(0, R, chunk):
            cr = min(chunk, R - i)
            us.append(torch.rand([batch_size * cr, n_importance]).reshape(batch_size, cr, n_importance))
        u = torch.cat(us, 1).to(rays_o.device)
    else:
        u = None
    ret = core.render(tp_input, None, z, rays_o, rays_d, near, far, tri_planes, n_importance, white_bkgd,
                      n_samples=n_samples, u=u)
    return [ret[k] for k in ret]


render_rays = render


def render_view(H, W, K, R, T, tri_planes, tp_input, renderer, n_samples=128, n_importance=128, white_bkgd=False, u=None,
                generator=None):
    """One full view without touching the host: rays + near/far by hl_camera_rays
    (SynBodyView_datasets.py:316-433 on the device), then the fused render.

    The reference builds the rays with numpy and uploads 6.3 MB per 512x512 view, and draws sample_pdf's uniforms on the
    CPU (renderer.py:545: 134 MB per view at 128 samples).  Here `u` may be a device tensor (H*W, n_importance); when
    None it is drawn on the device (`generator`: a device torch.Generator) - same distribution, different stream than the
    reference's CPU generator, so use render() with host-drawn u where bit-level replay of a reference run matters.
    Returns [rgb_map (H,W,3), acc_map (H,W), normal_map (H,W,3), depth_map (H,W)].
    """
    from ..SynBodyView_datasets import camera_rays
    core = renderer.module if hasattr(renderer, "module") else renderer
    dev = tri_planes.device
    bounds = tp_input["world_bounds"].reshape(-1, 2, 3)[0].detach().cpu().numpy()
    rays_o, rays_d, near, far, _ = camera_rays(H, W, K, R, T, bounds, dev, return_mask=False)
    if n_importance > 0 and u is None:
        u = torch.rand((H * W, n_importance), device=dev, generator=generator)
    ret = core.render(tp_input, None, None, rays_o[None], rays_d[None], near[None, :, None], far[None, :, None], tri_planes,
                      n_importance, white_bkgd, n_samples=n_samples, u=None if u is None else u.reshape(1, H * W, -1))
    out = [ret[k] for k in ret]
    return [out[0].reshape(H, W, 3), out[1].reshape(H, W), out[2].reshape(H, W, 3), out[3].reshape(H, W)]
