"""GPU parity of the renderer's training mode: forward with density noise and the HIP backward kernels against
gradient vectors from the reference's autograd (tests/golden/render_grad_*.npz) and against the oracle's autograd at a larger size.
Gradients are sums of ~1e4..1e6 fp32 terms accumulated in a different order than autograd's (MFMA chains, fixed-point tile sums on the
tri-plane, point ranges summed in a fixed order for the weight products): the bar is 2e-4 of the largest entry of each tensor (+2e-6 absolute),
written next to each check.  Since round 5 the backward has no floating-point atomics: two runs give the same bits (tested below)."""
import numpy as np
import pytest
import torch

from tests.test_oracle_render_grad import MLP_KEYS, load_grad_case, oracle_grads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def make_renderer(mlp, dev):
    from humanliff_amd.NeRF import Renderer
    r = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type='smpl', test=False)
    r.load_state_dict(mlp, strict=False)
    return r.to(dev)


def hip_grads(i, dev):
    r = make_renderer(i["mlp"], dev)
    planes = i["planes"].to(dev).clone().requires_grad_(True)          # (1,3,9,H,W)
    N = i["n_samples"]
    tp = {"world_bounds": i["bounds"][None].to(dev)}
    out = r.render(tp, None, i["z"][None].to(dev), i["rays_o"][None].to(dev), i["rays_d"][None].to(dev), i["near"][None, :, None].to(dev),
                   i["far"][None, :, None].to(dev), planes, N, i["white_bkgd"], u=i["u"].to(dev), noise=i["noise"].reshape(-1, 1).to(dev))
    assert out["rgb_map"].requires_grad and out["acc_map"].requires_grad and not out["depth_map"].requires_grad
    assert out["normal_map"] is out["rgb_map"]
    loss = (out["rgb_map"][0] * i["G_rgb"].to(dev)).sum() + (out["acc_map"][0] * i["G_acc"].to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    sd = dict(r.named_parameters())
    return out["rgb_map"][0].detach().cpu(), out["acc_map"][0].detach().cpu(), planes.grad[0].cpu(), {k: sd[k].grad.cpu() for k in MLP_KEYS}


def close(got, ref, what):
    err, scale = float((got - ref).abs().max()), float(ref.abs().max())
    # floor: the density deltas are differences of O(1) terms (cotangents ~1), so a sum of them carries ~1e-6 of absolute rounding
    assert err < 2e-6 + 2e-4 * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("name", ["a", "white"])
def test_gradients_match_reference_golden(name, dev):
    i, g = load_grad_case(name)
    rgb, acc, d_planes, d_mlp = hip_grads(i, dev)
    assert (rgb - torch.from_numpy(g["rgb"])).abs().max() < 2e-5
    assert (acc - torch.from_numpy(g["acc"])).abs().max() < 2e-5
    close(d_planes, torch.from_numpy(g["d_planes"]), "tri_planes")
    for k in MLP_KEYS:
        close(d_mlp[k], torch.from_numpy(g["d_" + k]), k)


def test_gradients_at_training_sample_counts(dev):
    """The fitting configuration's per-ray sizes - 128 stratified + 128 importance samples, 256x256x27 tri-plane - on 512 rays (what the
    oracle's autograd finishes in seconds); the gradient is a sum over rays, so this is the full-size arithmetic per ray."""
    from humanliff_amd import synthetic as syn
    g = torch.Generator().manual_seed(8)
    ro, rd, nr, fr = syn.orbit_rays(1, 8, 96, 96)
    pick = torch.nonzero(fr != 1).flatten()
    pick = pick[torch.randperm(pick.numel(), generator=g)[:512]]
    ro, rd, nr, fr = ro[pick], rd[pick], nr[pick], fr[pick]
    N = 128
    t = torch.linspace(0., 1., steps=N)
    z = nr[:, None] * (1. - t) + fr[:, None] * t
    mids = .5 * (z[:, 1:] + z[:, :-1])
    lower, upper = torch.cat([z[:, :1], mids], -1), torch.cat([mids, z[:, -1:]], -1)
    z = lower + (upper - lower) * torch.rand(z.shape, generator=g)
    i = dict(planes=syn.triplane(seed=13), bounds=torch.tensor(syn.WORLD_BOUNDS), mlp=syn.render_mlp_state(5),
             rays_o=ro, rays_d=rd, near=nr, far=fr, z=z, u=torch.rand((512, N), generator=g), noise=torch.randn((512, 2 * N), generator=g),
             G_rgb=torch.randn((512, 3), generator=g) / 512, G_acc=torch.randn((512,), generator=g) / 512, n_samples=N, white_bkgd=True)
    rgb, acc, d_planes, d_mlp = hip_grads(i, dev)
    o_rgb, o_acc, o_planes, o_mlp = oracle_grads(i)
    assert (rgb - o_rgb).abs().max() < 2e-5 and (acc - o_acc).abs().max() < 2e-5
    close(d_planes, o_planes, "tri_planes")
    for k in MLP_KEYS:
        close(d_mlp[k], o_mlp[k], k)


def test_gradients_are_the_same_bits_on_every_run(dev):
    """No floating-point atomics in the backward: the tri-plane gradient accumulates in 64-bit fixed point (integer additions commute), the
    weight gradients are partial results of fixed point ranges summed in a fixed order - five runs of the golden case (ragged sizes) and three
    at the fitting sample counts must agree bit for bit, images and all 15 gradient tensors."""
    from humanliff_amd import synthetic as syn
    i, _ = load_grad_case("a")
    g = torch.Generator().manual_seed(21)
    ro, rd, nr, fr = syn.orbit_rays(2, 8, 96, 96)
    pick = torch.nonzero(fr != 1).flatten()
    pick = pick[torch.randperm(pick.numel(), generator=g)[:2048]]
    ro, rd, nr, fr = ro[pick], rd[pick], nr[pick], fr[pick]
    N = 128
    t = torch.linspace(0., 1., steps=N)
    z = nr[:, None] * (1. - t) + fr[:, None] * t
    big = dict(planes=syn.triplane(seed=14), bounds=torch.tensor(syn.WORLD_BOUNDS), mlp=syn.render_mlp_state(6),
               rays_o=ro, rays_d=rd, near=nr, far=fr, z=z, u=torch.rand((2048, N), generator=g), noise=torch.randn((2048, 2 * N), generator=g),
               G_rgb=torch.randn((2048, 3), generator=g) / 2048, G_acc=torch.randn((2048,), generator=g) / 2048, n_samples=N, white_bkgd=False)
    for case, runs in ((i, 5), (big, 3)):
        ref = hip_grads(case, dev)
        assert float(ref[2].abs().max()) > 0 and all(float(v.abs().max()) > 0 for v in ref[3].values())
        for _ in range(runs - 1):
            got = hip_grads(case, dev)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
            assert torch.equal(got[2], ref[2]), f"tri-plane gradient differs by {float((got[2] - ref[2]).abs().max()):.3e}"
            for k in MLP_KEYS:
                assert torch.equal(got[3][k], ref[3][k]), f"{k} differs by {float((got[3][k] - ref[3][k]).abs().max()):.3e}"


def test_a_delta_that_is_not_finite_poisons_the_plane_gradient(dev):
    """The fixed-point tile sums cannot carry inf / NaN: a cotangent that is not finite makes the whole tri-plane gradient NaN (loud), where float
    atomics would have put NaN into the touched texels only."""
    i, _ = load_grad_case("a")
    i = dict(i)
    G = i["G_rgb"].clone()
    G[3, 1] = float("inf")
    i["G_rgb"] = G
    _, _, d_planes, _ = hip_grads(i, dev)
    assert bool(torch.isnan(d_planes).all())


def test_gradients_match_oracle_larger(dev):
    """1 000 rays x (48+48) samples on a 72x72 tri-plane (ragged scatter tiles), ragged last ray tile, stratified depths; cotangents
    like an MSE loss."""
    from humanliff_amd import synthetic as syn
    g = torch.Generator().manual_seed(3)
    ro, rd, nr, fr = syn.orbit_rays(5, 8, 64, 64)
    pick = torch.nonzero(fr != 1).flatten()
    pick = pick[torch.randperm(pick.numel(), generator=g)[:1000]]
    ro, rd, nr, fr = ro[pick], rd[pick], nr[pick], fr[pick]
    N = 48
    t = torch.linspace(0., 1., steps=N)
    z = nr[:, None] * (1. - t) + fr[:, None] * t
    mids = .5 * (z[:, 1:] + z[:, :-1])
    z = torch.cat([z[:, :1], mids], -1) + (torch.cat([mids, z[:, -1:]], -1) - torch.cat([z[:, :1], mids], -1)) * torch.rand(z.shape, generator=g)
    i = dict(planes=syn.triplane(seed=12, H=72, W=72), bounds=torch.tensor(syn.WORLD_BOUNDS), mlp=syn.render_mlp_state(4),
             rays_o=ro, rays_d=rd, near=nr, far=fr, z=z, u=torch.rand((1000, N), generator=g), noise=torch.randn((1000, 2 * N), generator=g),
             G_rgb=torch.randn((1000, 3), generator=g) / 1000, G_acc=torch.randn((1000,), generator=g) / 1000, n_samples=N, white_bkgd=False)
    rgb, acc, d_planes, d_mlp = hip_grads(i, dev)
    o_rgb, o_acc, o_planes, o_mlp = oracle_grads(i)
    assert (rgb - o_rgb).abs().max() < 2e-5 and (acc - o_acc).abs().max() < 2e-5
    close(d_planes, o_planes, "tri_planes")
    for k in MLP_KEYS:
        close(d_mlp[k], o_mlp[k], k)


@pytest.mark.parametrize("fused", [False, True])
def test_fitting_loop_two_subjects(dev, fused):
    """The shape of recon_NeRF/run_nerf_batch.py:236-265: tri_planes is a Parameter indexed per subject, batch of two subjects, MSE on
    rgb and acc, Adam on both parameter groups; the loss must go down and only the rendered subjects receive tri-plane gradient.
    fused=True: the fused optimizers do not bump Tensor._version; a version-keyed cache of the re-laid MLP once froze this loop."""
    from humanliff_amd import synthetic as syn
    torch.manual_seed(0)
    r = make_renderer(syn.render_mlp_state(3), dev)
    tri = torch.nn.Parameter((0.1 * torch.randn((3, 4, 3, 9, 32, 32))).to(dev))
    opt = torch.optim.Adam([{'params': list(r.parameters()), 'lr': 5e-4}, {'params': [tri], 'lr': 1e-2}], betas=(0.9, 0.999), fused=fused)
    ro, rd, nr, fr = syn.orbit_rays(2, 8, 32, 32)
    pick = torch.nonzero(fr != 1).flatten()[:256]
    ro, rd, nr, fr = (t[pick].to(dev) for t in (ro, rd, nr, fr))
    bs, R, N = 2, 256, 16
    target = torch.tensor([0.2, 0.5, 0.8], device=dev).expand(bs, R, 3)     # reachable: the loss can go to ~0
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].expand(bs, 2, 3).to(dev)}
    ids, layer = torch.tensor([0, 2]), torch.tensor([1, 3])
    losses = []
    for it in range(30):
        t = torch.linspace(0., 1., steps=N, device=dev)
        z = (nr[:, None] * (1. - t) + fr[:, None] * t)[None].expand(bs, R, N)
        out = r.render(tp, None, z, ro[None].expand(bs, R, 3), rd[None].expand(bs, R, 3), nr[None, :, None].expand(bs, R, 1),
                       fr[None, :, None].expand(bs, R, 1), tri[ids, layer], N, False)
        loss = ((out["rgb_map"] - target) ** 2).mean() + 0.1 * ((out["acc_map"] - 1.0) ** 2).mean()
        loss.backward()
        if it == 0:
            gsum = tri.grad.abs().sum(dim=(2, 3, 4, 5)).cpu()
            assert gsum[0, 1] > 0 and gsum[2, 3] > 0
            gsum[0, 1] = gsum[2, 3] = 0
            assert float(gsum.sum()) == 0.0
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < 0.5 * np.mean(losses[:5]), losses


def test_subject_streams_change_nothing(dev):
    """Renderer.subject_streams (on by default since round 5) puts subjects after the first on their own HIP streams (forward and, through autograd's stream
    rule, backward): three subjects with the switch on and off must give the same images and the same gradients, bit for bit (the backward has no
    floating-point atomics), repeated so that the allocator reuses blocks across streams."""
    from humanliff_amd import synthetic as syn
    torch.manual_seed(3)
    r = make_renderer(syn.render_mlp_state(3), dev)
    tri = torch.nn.Parameter((0.1 * torch.randn((3, 4, 3, 9, 32, 32))).to(dev))
    ro, rd, nr, fr = syn.orbit_rays(2, 8, 32, 32)
    pick = torch.nonzero(fr != 1).flatten()[:256]
    ro, rd, nr, fr = (t[pick].to(dev) for t in (ro, rd, nr, fr))
    bs, R, N = 3, 256, 16
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].expand(bs, 2, 3).to(dev)}
    ids, layer = torch.tensor([0, 2, 1], device=dev), torch.tensor([1, 3, 0], device=dev)
    t = torch.linspace(0., 1., steps=N, device=dev)
    z = (nr[:, None] * (1. - t) + fr[:, None] * t)[None].expand(bs, R, N).contiguous()
    g = torch.Generator().manual_seed(9)
    u = torch.rand((bs * R, N), generator=g).to(dev)
    noise = torch.randn((bs * R * 2 * N, 1), generator=g).to(dev)
    target = torch.rand((bs, R, 3), generator=g).to(dev)

    def run(on):
        r.subject_streams = on
        res = []
        for _ in range(4):
            r.zero_grad()
            tri.grad = None
            out = r.render(tp, None, z, ro[None].expand(bs, R, 3), rd[None].expand(bs, R, 3), nr[None, :, None].expand(bs, R, 1),
                           fr[None, :, None].expand(bs, R, 1), tri[ids, layer], N, False, u=u, noise=noise)
            loss = ((out["rgb_map"] - target) ** 2).mean() + 0.1 * ((out["acc_map"] - 1.0) ** 2).mean()
            loss.backward()
            res.append((out["rgb_map"].detach().clone(), out["acc_map"].detach().clone(), tri.grad.clone(),
                        [p.grad.clone() for p in r.parameters()]))
        torch.cuda.synchronize()
        return res
    ref = run(False)[0]
    for got in run(True):
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
        assert torch.equal(got[2], ref[2])
        for a, b in zip(got[3], ref[3]):
            assert torch.equal(a, b)
    r.subject_streams = True


def test_recon_twin_training_step(dev):
    """recon_NeRF/run_nerf_batch.py:236-265 through the mirrors of its own names: Renderer with the tri_planes Parameter inside,
    render(chunk, rays_o, ..., tp_input, renderer=DataParallel-like wrapper, perturb=1), TV + L1 regularisers, Adam on two groups."""
    import torch.nn.functional as F
    from humanliff_amd import synthetic as syn
    from humanliff_amd.recon_NeRF import Renderer, render
    torch.manual_seed(1)
    model = Renderer(use_canonical_space=False, num_instances=2, triplane_dim=32, triplane_ch=27, test=False)
    model.load_state_dict(syn.render_mlp_state(3), strict=False)
    model = model.to(dev)

    class Wrapped(torch.nn.Module):          # what nn.DataParallel / DDP present: the module under `.module`
        def __init__(self, m):
            super().__init__()
            self.module = m
    wrapped = Wrapped(model)
    grad_vars = [p for n, p in model.named_parameters() if n != 'tri_planes']
    opt = torch.optim.Adam([{'params': grad_vars, 'lr': 5e-4}, {'params': [model.tri_planes], 'lr': 1e-2}], betas=(0.9, 0.999))
    ro, rd, nr, fr = syn.orbit_rays(2, 8, 32, 32)
    pick = torch.nonzero(fr != 1).flatten()[:200]
    bs = 2
    ro, rd, nr, fr = (t[pick].to(dev)[None].expand(bs, *t[pick].shape) for t in (ro, rd, nr, fr))
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].expand(bs, 2, 3).to(dev), "instance_idx": torch.tensor([0, 1]),
          "cloth_layer_index": torch.tensor([2, 0])}
    target = torch.tensor([0.7, 0.3, 0.1], device=dev).expand(bs, 200, 3)
    losses = []
    for it in range(25):
        rgb, acc, _, _ = render(chunk=80000, rays_o=ro, rays_d=rd, tp_input=tp, near=nr, far=fr, perturb=1.0, n_samples=16, renderer=wrapped,
                                n_importance=16)
        tri = model.tri_planes[tp["instance_idx"], tp["cloth_layer_index"]]
        tv = F.l1_loss(tri[:, :, :, 0:-1, :], tri[:, :, :, 1:, :]) + F.l1_loss(tri[:, :, :, :, 0:-1], tri[:, :, :, :, 1:])
        loss = ((rgb - target) ** 2).mean() + 0.1 * ((acc - 1.0) ** 2).mean() + 0.01 * tv + 0.001 * tri.abs().mean()
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < 0.5 * np.mean(losses[:5]), losses
    untouched = model.tri_planes.detach()[0, 0]          # (instance 0, layer 0) was never rendered: only the L1/TV terms could move it - they did not see it
    assert torch.isfinite(untouched).all()


def test_large_ray_batches_are_split(dev, monkeypatch):
    """Ray batches beyond the 2 GiB matrix limit go down in pieces: forcing 64-ray pieces must reproduce the one-piece gradients."""
    from humanliff_amd.NeRF import Renderer
    i, _ = load_grad_case("a")
    ref = hip_grads(i, dev)
    monkeypatch.setattr(Renderer, "_train_ray_chunk", staticmethod(lambda S: 64))
    got = hip_grads(i, dev)
    assert (got[0] - ref[0]).abs().max() < 1e-6 and (got[1] - ref[1]).abs().max() < 1e-6
    close(got[2], ref[2], "tri_planes")
    for k in MLP_KEYS:
        close(got[3][k], ref[3][k], k)


def test_canonical_space_training_gradients_match_oracle(dev):
    """use_canonical_space=True with test=False (the TightCap fitting runs, README.md:123): sample points through the body deformation,
    density noise, gradients for the tri-plane and the MLP - against the oracle's deform_target2c + lookup + MLP + compositing under
    autograd (importance depths under no_grad) on a synthetic posed body."""
    from oracle import deform_oracle as do, render_oracle as orc
    from humanliff_amd import synthetic as syn
    from humanliff_amd.NeRF import Renderer
    V, H, W, N = 1500, 24, 24, 24
    g = torch.Generator().manual_seed(5)
    cpu_model = syn.smpl_like_model(V, 7)
    pose = syn.smpl_like_pose(V, cpu_model, 17, n_points=8)
    mlp = syn.render_mlp_state(3)
    r = Renderer(use_canonical_space=True, triplane_dim=64, triplane_ch=27, test=False)
    r.load_state_dict(mlp, strict=False)
    r = r.to(dev)
    r.SMPL_NEUTRAL = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in cpu_model.items()}
    planes = syn.triplane(seed=11, H=64, W=64)
    centre = pose["vertices"][0].mean(0)
    lo, hi = pose["vertices"][0].min(0).values - 0.1, pose["vertices"][0].max(0).values + 0.1
    ro, rd, _, _ = syn.orbit_rays(4, 36, H, W)
    ro = ro + centre
    nr, fr = syn.near_far_from_bounds(torch.stack([lo, hi]).double().numpy(), ro.double().numpy(), rd.double().numpy())
    nr, fr = torch.from_numpy(nr).float(), torch.from_numpy(fr).float()
    keep = torch.nonzero(fr != 1).flatten()[:300]            # rays through the box (the fitting loop samples inside the mask's box)
    ro, rd, nr, fr = ro[keep], rd[keep], nr[keep], fr[keep]
    R = ro.shape[0]
    t = torch.linspace(0.0, 1.0, steps=N)
    z = nr[:, None] * (1.0 - t) + fr[:, None] * t
    u = torch.rand((R, N), generator=g)
    noise = torch.randn((R, 2 * N), generator=g)
    G_rgb, G_acc = torch.randn((R, 3), generator=g) / R, torch.randn((R,), generator=g) / R

    tri = planes.to(dev).clone().requires_grad_(True)
    out = r.render(pose, None, z[None].to(dev), ro[None].to(dev), rd[None].to(dev), nr[None, :, None].to(dev), fr[None, :, None].to(dev), tri,
                   N, False, u=u.to(dev), noise=noise.reshape(-1, 1).to(dev))
    ((out["rgb_map"][0] * G_rgb.to(dev)).sum() + (out["acc_map"][0] * G_acc.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    sd = dict(r.named_parameters())
    # the same step again: the same bits (k_plane_scatter_pts accumulates in fixed point, k_wgrad_finish sums in a fixed order)
    first = [tri.grad.clone()] + [sd[k].grad.clone() for k in MLP_KEYS]
    tri.grad = None
    r.zero_grad()
    out2 = r.render(pose, None, z[None].to(dev), ro[None].to(dev), rd[None].to(dev), nr[None, :, None].to(dev), fr[None, :, None].to(dev), tri,
                    N, False, u=u.to(dev), noise=noise.reshape(-1, 1).to(dev))
    ((out2["rgb_map"][0] * G_rgb.to(dev)).sum() + (out2["acc_map"][0] * G_acc.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    for a, b in zip(first, [tri.grad] + [sd[k].grad for k in MLP_KEYS]):
        assert torch.equal(a, b)

    # oracle
    tb = pose["t_world_bounds"][0]
    vd = rd / rd.norm(dim=1, keepdim=True)
    op = planes[0].clone().requires_grad_(True)
    om = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}

    def evaluate(zz):
        S = zz.shape[1]
        pts = (ro[:, None, :] + rd[:, None, :] * zz[:, :, None]).reshape(-1, 3)
        with torch.no_grad():
            can, cd, _ = do.deform_target2c(cpu_model, pose, pts, vd[:, None, :].expand(R, S, 3).reshape(-1, 3))
        rgb_raw, sig = orc.mlp(om, orc.plane_features(op, can, tb), cd)
        return rgb_raw.reshape(R, S, 3), sig.reshape(R, S)

    with torch.no_grad():
        _, sig_c = evaluate(z)
        z_all = orc.importance_z(sig_c, z, rd, u)
    rgb_raw, sig = evaluate(z_all)
    rgb, acc, _ = orc.composite(rgb_raw, sig, z_all, False, noise)
    ((rgb * G_rgb).sum() + (acc * G_acc).sum()).backward()
    assert (out["rgb_map"][0].detach().cpu() - rgb.detach()).abs().max() < 5e-5
    # a sample point whose nearest vertex is a near-tie may deform through the other vertex in float32 (see the inference test): allow
    # those few points' contribution
    err = (tri.grad[0].cpu() - op.grad).abs().max()
    assert err < 2e-6 + 2e-3 * op.grad.abs().max(), f"tri_planes {float(err):.3e} vs {float(op.grad.abs().max()):.3e}"
    for k in MLP_KEYS:
        e, sc = float((sd[k].grad.cpu() - om[k].grad).abs().max()), float(om[k].grad.abs().max())
        assert e < 2e-6 + 2e-3 * sc, f"{k}: {e:.3e} vs {sc:.3e}"


def test_training_mode_draws_the_cpu_generators_uniforms_on_the_device(dev):
    """u = None in training mode (renderer.py:545: torch.rand on the CPU generator): from 65 536 numbers on they are written by the device
    (hl_mt19937_uniform continues the generator's stream) - same images, same gradients, and the CPU generator ends where torch.rand leaves it."""
    from humanliff_amd import synthetic as syn
    r = make_renderer(syn.render_mlp_state(2), dev)
    bs, R, N = 2, 640, 64                                       # 81 920 uniforms
    ro, rd, nr, fr = syn.orbit_rays(3, 8, 64, 64)
    pick = torch.nonzero(fr != 1).flatten()[:R]
    ro, rd, nr, fr = (t[pick].to(dev) for t in (ro, rd, nr, fr))
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].expand(bs, 2, 3).to(dev)}
    t = torch.linspace(0., 1., steps=N, device=dev)
    z = (nr[:, None] * (1. - t) + fr[:, None] * t)[None].expand(bs, R, N).contiguous()
    g = torch.Generator().manual_seed(4)
    noise = torch.randn((bs * R * 2 * N, 1), generator=g).to(dev)
    planes = torch.stack([syn.triplane(seed=3, H=64, W=64)[0], syn.triplane(seed=4, H=64, W=64)[0]]).to(dev)

    def run(u):
        tri = planes.clone().requires_grad_(True)
        r.zero_grad()
        out = r.render(tp, None, z, ro[None].expand(bs, R, 3), rd[None].expand(bs, R, 3), nr[None, :, None].expand(bs, R, 1),
                       fr[None, :, None].expand(bs, R, 1), tri, N, False, u=u, noise=noise)
        (out["rgb_map"].square().mean() + out["acc_map"].mean()).backward()
        torch.cuda.synchronize()
        return out["rgb_map"].detach().clone(), tri.grad.clone(), [p.grad.clone() for p in r.parameters()]
    torch.manual_seed(77)
    torch.rand(100)                                             # (the generator mid-block)
    a = run(None)
    after_a = torch.rand(5)
    torch.manual_seed(77)
    torch.rand(100)
    u = torch.rand((bs * R, N))
    b = run(u.to(dev))
    after_b = torch.rand(5)
    assert torch.equal(after_a, after_b)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        assert torch.equal(x, y)
