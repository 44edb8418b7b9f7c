"""GPU parity: HIP ray-march path (through the C ABI) vs golden vectors from the reference and
vs the CPU oracle on the same seeded inputs.  Tolerances are fp32-roundoff class and written
next to each check."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests.golden_util import load_render_case, psnr

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def make_renderer(mlp, dev):
    from humanliff_amd.NeRF import Renderer
    r = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type='smpl', test=True)
    missing = r.load_state_dict(mlp, strict=False)
    assert set(missing.missing_keys) <= {"view_enc._freqs", "view_enc._phases"}
    return r.to(dev)


PRODUCTS = ["fp16x2", "bf16x3", "fp32"]   # Renderer.mlp_products: two fp16 planes / three partial products (k_march_plw<2>), exact three-way bf16 splits / six products (k_march_plw<3>), the fp32-MFMA kernel


def hip_render(i, dev, z_vals=None, mlp_fp16=False, products=None):
    r = make_renderer(i["mlp"], dev)
    r.mlp_fp16 = mlp_fp16
    if products is not None:
        r.mlp_products = products
    tp = {"world_bounds": i["bounds"][None].to(dev)}
    out = r.render(tp, None, z_vals, i["rays_o"][None].to(dev), i["rays_d"][None].to(dev), i["near"][None, :, None].to(dev),
                   i["far"][None, :, None].to(dev), i["planes"].to(dev), i["n_importance"], i["white_bkgd"],
                   n_samples=i["n_samples"], u=i["u"].to(dev))
    torch.cuda.synchronize()
    return r, {k: v[0].cpu() for k, v in out.items()}


@pytest.mark.parametrize("products", PRODUCTS)
@pytest.mark.parametrize("name", ["a", "b", "c", "d", "e", "f"])
def test_render_matches_reference_golden(name, products, dev):
    """All three product modes against the REFERENCE's renders with the same bounds (fp16x2: two fp16 planes per operand, 2^-20; bf16x3: exact splits, dropped terms below one fp32 rounding).
    Cases d, e, f (round 6): the MLP's layers 2^-8 below / 2^4 above nn.Linear's initialisation (syn.LAYER_EXP) - the range of the weights, which the scaled fp16 planes
    (k_mlp_scales_h2) must not care about; d also reaches F.softplus's x > 20 branch (pre-activations ~100: the log2-domain softplus's med3)."""
    i, e = load_render_case(name)
    r, out = hip_render(i, dev, products=products)
    assert r.mlp_products == products
    # intermediates kept in the workspace: the raw (sigma, r, g, b) records of the coarse points come first, tile-major
    R, N = i["rays_o"].shape[0], i["n_samples"]
    tiles = (R + 31) // 32
    rec = r._ws.cpu()[:tiles * N * 32 * 4].reshape(tiles, N, 32, 4)
    sigma = rec[..., 0].permute(0, 2, 1).reshape(tiles * 32, N)[:R]
    # raw densities: 2e-5 while |sigma| <= 4 (cases a, b, c, e, f), relative beyond (case d, syn.LAYER_EXP: |sigma| up to 12.9 - measured 2.1e-5 / 2.0e-5 / 1.3e-5 in the three modes)
    assert (sigma - e["sigma_coarse"]).abs().max() < 5e-6 * max(4.0, float(e["sigma_coarse"].abs().max()))
    assert (out["rgb_map"] - e["rgb"]).abs().max() < 2e-5          # colours in [0,1]
    assert (out["acc_map"] - e["acc"]).abs().max() < 2e-5
    assert (out["depth_map"] - e["depth"]).abs().max() < 5e-5
    assert psnr(out["rgb_map"], e["rgb"]) > 90.0                   # north-star bar is 45 dB
    assert out["normal_map"].data_ptr() == out["rgb_map"].data_ptr() or torch.equal(out["normal_map"], out["rgb_map"])


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_fp16_mlp_mode_meets_the_north_star_bar_against_the_reference(name, dev):
    """Renderer.mlp_fp16 (opt-in; k_march16: fp16 operands, fp32 accumulation, everything around the MLP fp32) against the REFERENCE's golden
    renders: the north-star bar is PSNR >= 45 dB."""
    i, e = load_render_case(name)
    _, ref32 = hip_render(i, dev, products="fp32")
    _, out = hip_render(i, dev, mlp_fp16=True)
    assert torch.isfinite(out["rgb_map"]).all() and not torch.equal(out["rgb_map"], ref32["rgb_map"])     # the mode really switched arithmetic
    p_ref, p_32 = psnr(out["rgb_map"], e["rgb"]), psnr(out["rgb_map"], ref32["rgb_map"])
    print(f"fp16 MLP, case {name}: PSNR {p_ref:.1f} dB against the reference's render, {p_32:.1f} dB against the fp32 mode; "
          f"max-abs rgb {float((out['rgb_map'] - e['rgb']).abs().max()):.2e}, acc {float((out['acc_map'] - e['acc']).abs().max()):.2e}")
    assert p_ref > 60.0 and p_32 > 60.0
    assert (out["acc_map"] - e["acc"]).abs().max() < 5e-3 and (out["depth_map"] - e["depth"]).abs().max() < 5e-3


def test_importance_stage_matches_oracle(dev):
    """k_importance alone, fed the oracle's coarse densities: merged depths must agree to 1e-5 of the
    ray span and be sorted."""
    from oracle import render_oracle as ro
    from humanliff_amd import _lib
    i, e = load_render_case("c")
    R, N = i["rays_o"].shape[0], i["n_samples"]
    t = torch.linspace(0, 1, N)
    z = i["near"][:, None] * (1 - t) + i["far"][:, None] * t
    want = ro.importance_z(e["sigma_coarse"], z, i["rays_d"], i["u"])
    from humanliff_amd.NeRF.renderer import tile_rows, untile_rows
    L = _lib.lib()
    tiles = (R + 31) // 32
    z_all = torch.empty(tiles * 32 * 2 * N, device=dev)
    d = lambda x: x.contiguous().to(dev)  # noqa: E731
    sig, rd, nr, fr, u = d(tile_rows(e["sigma_coarse"])), d(i["rays_d"]), d(i["near"]), d(i["far"]), d(i["u"])
    _lib.check(L.hl_render_importance(_lib.ptr(sig), _lib.ptr(rd), _lib.ptr(nr), _lib.ptr(fr), None, _lib.ptr(u),
                                      R, N, N, _lib.ptr(z_all), _lib.stream_ptr()))
    torch.cuda.synchronize()
    got = untile_rows(z_all.cpu(), R, 2 * N)
    assert (got[:, 1:] >= got[:, :-1]).all()
    span = (i["far"] - i["near"])[:, None]
    assert ((got - want).abs() / span).max() < 1e-4
    # explicit z_vals input == generated linspace
    zd = d(z)
    z_all2 = torch.empty_like(z_all)
    _lib.check(L.hl_render_importance(_lib.ptr(sig), _lib.ptr(rd), _lib.ptr(nr), _lib.ptr(fr), _lib.ptr(zd), _lib.ptr(u),
                                      R, N, N, _lib.ptr(z_all2), _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert (untile_rows(z_all2.cpu(), R, 2 * N) - got).abs().max() < 1e-6


def test_no_importance_and_explicit_z(dev):
    """n_importance=0 path and caller-supplied (perturbed) z_vals vs the oracle."""
    from oracle import render_oracle as ro
    i, _ = load_render_case("a")
    i = dict(i)
    i["n_importance"] = 0
    R, N = i["rays_o"].shape[0], i["n_samples"]
    g = torch.Generator().manual_seed(77)
    t = torch.linspace(0, 1, N)
    z = i["near"][:, None] * (1 - t) + i["far"][:, None] * t
    z = z + (torch.rand((R, N), generator=g) - 0.5) * (i["far"] - i["near"])[:, None] / (2 * N)
    _, out = hip_render(i, dev, z_vals=z[None].to(dev))
    rgb, acc, depth = ro.render_rays(i["mlp"], i["planes"][0], i["bounds"], i["rays_o"], i["rays_d"], i["near"], i["far"],
                                     N, 0, z_vals=z)
    assert (out["rgb_map"] - rgb).abs().max() < 2e-5
    assert (out["acc_map"] - acc).abs().max() < 2e-5
    assert (out["depth_map"] - depth).abs().max() < 5e-5


def test_render_function_chunks_like_reference(dev):
    """render() (the 'render_rays' of BASELINE.json): list output order, ragged last chunk."""
    from humanliff_amd.NeRF import render
    i, e = load_render_case("a")
    r = make_renderer(i["mlp"], dev)
    tp = {"world_bounds": i["bounds"][None].to(dev)}
    torch.manual_seed(5)  # the reference draws u per chunk from the CPU generator
    ret = render(chunk=100, rays_o=i["rays_o"][None].to(dev), rays_d=i["rays_d"][None].to(dev),
                 near=i["near"][None].to(dev), far=i["far"][None].to(dev), tri_planes=i["planes"].to(dev), tp_input=tp,
                 renderer=r, n_samples=i["n_samples"], perturb=0., n_importance=i["n_importance"])
    assert len(ret) == 4
    rgb, acc, normal, depth = [x.cpu() for x in ret]
    assert rgb.shape == (1, 256, 3) and acc.shape == (1, 256) and depth.shape == (1, 256)
    assert torch.equal(rgb, normal)
    # chunked u draws differ from the single-draw fixture, so compare against the oracle run chunk-wise
    from oracle import render_oracle as ro
    torch.manual_seed(5)
    want = []
    for s in range(0, 256, 100):
        sl = slice(s, min(s + 100, 256))
        u = torch.rand([sl.stop - sl.start, i["n_importance"]])
        want.append(ro.render_rays(i["mlp"], i["planes"][0], i["bounds"], i["rays_o"][sl], i["rays_d"][sl], i["near"][sl],
                                   i["far"][sl], i["n_samples"], i["n_importance"], u=u)[0])
    assert (rgb[0] - torch.cat(want)).abs().max() < 2e-5


def test_production_shape_vs_oracle(dev):
    """256x256 planes, 128+128 samples, 1500 rays of a 512x512 view (ragged vs the 256-ray workgroup)."""
    from oracle import render_oracle as ro
    from humanliff_amd import synthetic as syn
    planes = syn.triplane(seed=11, H=256, W=256)
    mlp = syn.render_mlp_state(3, gain=2.0)
    ro_, rd_, nr_, fr_ = syn.orbit_rays(7, 36, 512, 512)
    sl = slice(512 * 250 + 11, 512 * 250 + 11 + 1500)
    i = dict(planes=planes, bounds=torch.tensor(syn.WORLD_BOUNDS), rays_o=ro_[sl], rays_d=rd_[sl], near=nr_[sl], far=fr_[sl],
             u=syn.importance_u(1500, 128, seed=5), mlp=mlp, n_samples=128, n_importance=128, white_bkgd=False)
    _, out = hip_render(i, dev)
    rgb, acc, depth = ro.render_rays(mlp, planes[0], i["bounds"], i["rays_o"], i["rays_d"], i["near"], i["far"], 128, 128,
                                     u=i["u"])
    assert (out["rgb_map"] - rgb).abs().max() < 5e-5
    assert (out["acc_map"] - acc).abs().max() < 5e-5
    assert (out["depth_map"] - depth).abs().max() < 2e-4
    assert psnr(out["rgb_map"], rgb) > 80.0


@pytest.mark.parametrize("products", PRODUCTS)
def test_sample_counts_beyond_128_match_oracle(products, dev):
    """192 + 192 samples per ray: more than two samples per lane in k_importance (the densities are then read per ray instead of being
    fetched up front), three 64-sample groups in the scans and a 256-slot sort - against the oracle on 300 rays."""
    from oracle import render_oracle as ro
    from humanliff_amd import synthetic as syn
    planes = syn.triplane(seed=13, H=64, W=64)
    mlp = syn.render_mlp_state(3, gain=2.0)
    ro_, rd_, nr_, fr_ = syn.orbit_rays(3, 36, 64, 64)
    sl = slice(64 * 20 + 5, 64 * 20 + 5 + 300)
    N = 192
    i = dict(planes=planes, bounds=torch.tensor(syn.WORLD_BOUNDS), rays_o=ro_[sl], rays_d=rd_[sl], near=nr_[sl], far=fr_[sl],
             u=syn.importance_u(300, N, seed=6), mlp=mlp, n_samples=N, n_importance=N, white_bkgd=True)
    _, out = hip_render(i, dev, products=products)
    rgb, acc, depth = ro.render_rays(mlp, planes[0], i["bounds"], i["rays_o"], i["rays_d"], i["near"], i["far"], N, N, u=i["u"], white_bkgd=True)
    assert (out["rgb_map"] - rgb).abs().max() < 5e-5
    assert (out["acc_map"] - acc).abs().max() < 5e-5
    assert (out["depth_map"] - depth).abs().max() < 2e-4


def test_bad_arguments_raise(dev):
    from humanliff_amd.NeRF import Renderer
    i, _ = load_render_case("a")
    r = make_renderer(i["mlp"], dev)
    tp = {"world_bounds": i["bounds"][None].to(dev)}
    with pytest.raises(AssertionError):   # n_samples != n_importance (renderer.py:250)
        r.render(tp, None, None, i["rays_o"][None].to(dev), i["rays_d"][None].to(dev), i["near"][None].to(dev),
                 i["far"][None].to(dev), i["planes"].to(dev), 16, False, n_samples=32)
    with pytest.raises(RuntimeError):     # canonical space needs the body model (a licensed asset the caller provides)
        Renderer(use_canonical_space=True, triplane_ch=27).render(tp, None, None, None, None, None, None, i["planes"])


def test_density_grid_matches_oracle(dev):
    """extract_geometry's density lattice on the coarse kernel vs the oracle MLP at the
    reference's meshgrid points (renderer.py:297-318); ragged resolution."""
    from oracle import render_oracle as ro
    from humanliff_amd import synthetic as syn
    planes = syn.triplane(seed=11, H=64, W=64)
    mlp = syn.render_mlp_state(3, gain=2.0)
    r = make_renderer(mlp, dev)
    bounds = torch.tensor(syn.WORLD_BOUNDS)
    N = 21
    u = r.density_grid({"world_bounds": bounds[None].to(dev)}, planes.to(dev), resolution=N, rays_per_launch=200).cpu()
    X, Y, Z = [torch.linspace(float(bounds[0, k]), float(bounds[1, k]), N) for k in range(3)]
    xx, yy, zz = torch.meshgrid(X, Y, Z, indexing="ij")
    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=1)
    want = -ro.mlp(mlp, ro.plane_features(planes[0], pts, bounds)).reshape(N, N, N)
    assert u.shape == (N, N, N)
    assert (u - want).abs().max() < 2e-5
    with pytest.raises(ImportError):
        r.extract_geometry({"world_bounds": bounds[None].to(dev)}, planes.to(dev), resolution=8)


@pytest.mark.parametrize("products", PRODUCTS)
@pytest.mark.parametrize("R", [1, 31, 33, 64, 127, 129, 255, 257])
def test_ragged_ray_counts(R, products, dev):
    """Ray counts that leave most of the last 256-ray workgroup / 32-ray tile empty (workspace is sized for
    ceil(R/32) tiles only; idle waves must not touch memory beyond it)."""
    from oracle import render_oracle as ro
    i, _ = load_render_case("a")
    i = dict(i)
    for k in ("rays_o", "rays_d", "near", "far", "u"):
        i[k] = i[k][:R] if R <= 256 else torch.cat([i[k], i[k][:R - 256]])
    _, out = hip_render(i, dev, products=products)
    rgb, acc, depth = ro.render_rays(i["mlp"], i["planes"][0], i["bounds"], i["rays_o"], i["rays_d"], i["near"], i["far"],
                                     i["n_samples"], i["n_importance"], u=i["u"])
    assert out["rgb_map"].shape == (R, 3)
    assert (out["rgb_map"] - rgb).abs().max() < 2e-5
    assert (out["depth_map"] - depth).abs().max() < 5e-5


@pytest.mark.parametrize("R", [1, 33, 127, 129, 257])
def test_ragged_ray_counts_fp16_mlp(R, dev):
    """The same for the opt-in fp16-operand MLP (k_march16: 128-ray workgroups): ragged tiles / workgroups, tolerance of the mode."""
    from oracle import render_oracle as ro
    i, _ = load_render_case("a")
    i = dict(i)
    for k in ("rays_o", "rays_d", "near", "far", "u"):
        i[k] = i[k][:R] if R <= 256 else torch.cat([i[k], i[k][:R - 256]])
    _, out = hip_render(i, dev, mlp_fp16=True)
    rgb, acc, depth = ro.render_rays(i["mlp"], i["planes"][0], i["bounds"], i["rays_o"], i["rays_d"], i["near"], i["far"],
                                     i["n_samples"], i["n_importance"], u=i["u"])
    assert out["rgb_map"].shape == (R, 3) and torch.isfinite(out["rgb_map"]).all()
    assert (out["rgb_map"] - rgb).abs().max() < 2e-3 and (out["depth_map"] - depth).abs().max() < 5e-3


# ---- per-view ray generation on the device ---------------------------------------------------
def _cam_cases():
    import numpy as np
    g = np.load(os.path.join(GOLDEN, "camera_rays.npz"))
    return [str(n) for n in g["names"]]


@pytest.mark.parametrize("name", _cam_cases())
def test_camera_rays_match_reference_golden(name, dev):
    """hl_camera_rays against get_rays / get_near_far of the reference (float64 arithmetic rounded to float32: equal
    up to one float32 ulp where the summation order of a 3-term dot differs)."""
    import numpy as np
    from humanliff_amd.SynBodyView_datasets import camera_rays
    g = np.load(os.path.join(GOLDEN, "camera_rays.npz"))
    H, W = [int(v) for v in g[f"{name}_HW"]]
    ro, rd, near, far, mask = camera_rays(H, W, g[f"{name}_K"], g[f"{name}_R"], g[f"{name}_T"], g[f"{name}_bounds"], dev)
    assert np.array_equal(mask.cpu().numpy(), g[f"{name}_mask"])
    for got, want in [(ro, g[f"{name}_rays_o"]), (rd, g[f"{name}_rays_d"]), (near, g[f"{name}_near"]), (far, g[f"{name}_far"])]:
        got = got.cpu().numpy()
        ulp = np.spacing(np.abs(want).astype(np.float32))
        assert (np.abs(got.astype(np.float64) - want.astype(np.float64)) <= ulp).all()
        assert (got != want).mean() < 0.01


def test_camera_rays_fullsize_match_oracle(dev):
    """512x512 view: device rays equal the oracle's."""
    import numpy as np
    from oracle import camera_oracle as co
    from humanliff_amd import synthetic as syn
    from humanliff_amd.SynBodyView_datasets import camera_rays, get_rays
    H = W = 512
    K, c2w, cam = syn.orbit_camera(7, 36, H, W)
    R = c2w.T.copy()
    T = (-R @ cam).reshape(3, 1)
    b = np.asarray(syn.WORLD_BOUNDS, dtype=np.float32)
    ro, rd, near, far, mask = camera_rays(H, W, K, R, T, b, dev)
    oro, ord_, onear, ofar, omask = co.camera_rays(H, W, K, R, T, b)
    assert np.array_equal(mask.cpu().numpy(), omask)
    for got, want in [(ro, oro), (rd, ord_), (near, onear), (far, ofar)]:
        got = got.cpu().numpy()
        ulp = np.spacing(np.abs(want))
        assert (np.abs(got.astype(np.float64) - want.astype(np.float64)) <= ulp).all()
        assert (got != want).mean() < 1e-3
    ro2, rd2 = get_rays(H, W, K, R, T, dev)
    assert ro2.shape == (H, W, 3) and torch.equal(rd2.reshape(-1, 3), rd)


def test_render_view_equals_render_on_host_made_rays(dev):
    """render_view (device rays + the fused render) == render() fed with the oracle's host-made rays and the same u."""
    import numpy as np
    from oracle import camera_oracle as co
    from humanliff_amd import synthetic as syn
    from humanliff_amd.NeRF import render_view
    H = W = 96
    K, c2w, cam = syn.orbit_camera(3, 36, H, W)
    R = c2w.T.copy()
    T = (-R @ cam).reshape(3, 1)
    b = np.asarray(syn.WORLD_BOUNDS, dtype=np.float32)
    r = make_renderer(syn.render_mlp_state(3), dev)
    planes = syn.triplane(seed=11).to(dev)
    tp = {"world_bounds": torch.from_numpy(b)[None].to(dev)}
    u = torch.rand((H * W, 32), generator=torch.Generator().manual_seed(5)).to(dev)
    got = render_view(H, W, K, R, T, planes, tp, r, n_samples=32, n_importance=32, u=u)
    oro, ord_, onear, ofar, _ = co.camera_rays(H, W, K, R, T, b)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    want = r.render(tp, None, None, t(oro)[None], t(ord_)[None], t(onear)[None], t(ofar)[None], planes, 32, False, n_samples=32,
                    u=u[None])
    assert got[0].shape == (H, W, 3) and got[3].shape == (H, W)
    # identical kernels on rays that agree to <= 1 ulp on <0.1 % of the elements
    assert (got[0].reshape(-1, 3) - want["rgb_map"][0]).abs().max() < 1e-5
    assert (got[3].reshape(-1) - want["depth_map"][0]).abs().max() < 1e-5


@pytest.mark.parametrize("R,N,white", [(1000, 32, False), (4096, 64, True), (777, 128, False)])
def test_evaluate_once_pipeline_is_bit_identical_to_reevaluation(dev, R, N, white):
    """Default hl_render_rays evaluates every sample point once (coarse points in pass A, importance points in pass B) and
    merges; HL_RENDER_REEVALUATE runs the fine pass over all points like the reference.  Same values, same order: equal bits."""
    from humanliff_amd import synthetic as syn
    r = make_renderer(syn.render_mlp_state(3), dev)
    r.mlp_products = "fp32"                      # the reference's schedule exists on the fp32-MFMA kernel only: the identity is a statement about that kernel
    planes = syn.triplane(seed=11).to(dev)
    ro, rd, nr, fr = [t[:R].to(dev) for t in syn.orbit_rays(5, 36, 64, 64)] if R <= 4096 else None
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
    u = torch.rand((R, N), generator=torch.Generator().manual_seed(R)).to(dev)
    a = r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, N, white, n_samples=N, u=u[None])
    a = {k: v.clone() for k, v in a.items()}
    b = r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, N, white, n_samples=N, u=u[None], reevaluate=True)
    for k in ("rgb_map", "acc_map", "depth_map"):
        assert torch.equal(a[k], b[k]), k
    assert float(a["acc_map"].max()) > 0.05      # not a vacuous all-empty render
    # the split-product modes against the same schedule: not the same bits (three fp16 / six bf16 partial products per fp32 product), the same image
    for mode in ("fp16x2", "bf16x3"):
        r.mlp_products = mode
        c = r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, N, white, n_samples=N, u=u[None])
        assert not torch.equal(c["rgb_map"], b["rgb_map"])
        assert (c["rgb_map"] - b["rgb_map"]).abs().max() < 5e-6 and (c["acc_map"] - b["acc_map"]).abs().max() < 5e-6, mode
        assert (c["depth_map"] - b["depth_map"]).abs().max() < 2e-5, mode
    from humanliff_amd.NeRF import Renderer
    assert Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type='smpl', test=True).mlp_products == "fp16x2"   # the default


@pytest.mark.parametrize("R,N,white", [(1, 32, False), (33, 32, True), (1000, 64, False), (4096, 128, True), (777, 128, False), (2049, 128, False), (300, 3, False)])
def test_one_pass_fine_launch_is_bit_identical_to_the_four_launch_pipeline(dev, R, N, white):
    """Round 6 (north star: sample + MLP + alpha-composite in ONE pass): the default schedule of the fp16x2 mode is two launches - the coarse evaluate, then
    k_march_plw<2, false, true>, whose waves draw the importance depths of their 32 rays from the coarse records (up_sample / sample_pdf, renderer.py:158-170, 533-563),
    evaluate them and composite coarse + new samples in depth order as they go (renderer.py:172-231, 252-253) - no k_importance, no fine records, no k_composite.
    Renderer.four_launch keeps rounds 2-5's evaluate / k_importance / evaluate / k_composite schedule: same depths, same records, same compositing order and
    arithmetic - the images must be the same BITS, for ragged ray counts (tiles and workgroups mostly empty), every sample count the fused launch takes (3 ... 128),
    white background, and repeated calls."""
    from humanliff_amd import synthetic as syn
    r = make_renderer(syn.render_mlp_state(3), dev)
    assert r.mlp_products == "fp16x2"
    planes = syn.triplane(seed=11).to(dev)
    ro, rd, nr, fr = [t[:R].to(dev) for t in syn.orbit_rays(5, 36, 64, 64)]
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
    u = torch.rand((R, N), generator=torch.Generator().manual_seed(R + N)).to(dev)
    outs = []
    for four in (True, False, False):
        r.four_launch = four
        o = r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, N, white, n_samples=N, u=u[None])
        outs.append({k: o[k].clone() for k in ("rgb_map", "acc_map", "depth_map")})
    for k in ("rgb_map", "acc_map", "depth_map"):
        assert bool(torch.isfinite(outs[1][k]).all()), k
        assert torch.equal(outs[0][k], outs[1][k]), (k, float((outs[0][k] - outs[1][k]).abs().max()))
        assert torch.equal(outs[1][k], outs[2][k]), k
    if R >= 100:
        assert float(outs[1]["acc_map"].max()) > 0.05      # not a vacuous all-empty render
    # the schedules that have no one-pass form fall back to the four launches by themselves: caller-given depths, the other product modes
    z = (nr[:, None] * (1 - torch.linspace(0, 1, N, device=dev)) + fr[:, None] * torch.linspace(0, 1, N, device=dev)) * (1 + 1e-3 * torch.rand((R, N), device=dev)).sort(dim=1)[0]
    r.four_launch = False
    a = r.render(tp, None, z[None], ro[None], rd[None], nr[None], fr[None], planes, N, white, u=u[None])
    r.four_launch = True
    b = r.render(tp, None, z[None], ro[None], rd[None], nr[None], fr[None], planes, N, white, u=u[None])
    assert torch.equal(a["rgb_map"], b["rgb_map"])


# ---- canonical-space deformation ----------------------------------------------------------------
@pytest.mark.parametrize("name", ["small", "mid"])
def test_deform_matches_reference_golden(name, dev):
    """hl_deform_points (1-NN + folded inverse / forward LBS) against the reference's deform_target2c on the synthetic body."""
    import numpy as np
    from humanliff_amd import synthetic as syn
    from humanliff_amd.NeRF.deform import deform_target2c
    g = np.load(os.path.join(GOLDEN, "deform.npz"))
    V, P, seed = [int(v) for v in g[f"{name}_VP"]]
    model = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.smpl_like_model(V, seed).items()}
    pose = syn.smpl_like_pose(V, model={k: v.cpu() if torch.is_tensor(v) else v for k, v in model.items()}, seed=seed + 10, n_points=P)
    can, cd, box = deform_target2c(model, pose, pose["pts"].to(dev), pose["viewdirs"].to(dev))
    assert np.abs(can.cpu().numpy() - g[f"{name}_can_pts"]).max() < 2e-5      # float32 chains of ~30 ops on O(1) values
    assert np.abs(cd.cpu().numpy() - g[f"{name}_can_dirs"]).max() < 2e-5
    assert torch.equal(box, pose["t_world_bounds"])
    can2, none, _ = deform_target2c(model, pose, pose["pts"].to(dev))
    assert none is None and torch.equal(can, can2)


def test_deform_fullsize_matches_oracle(dev):
    """6 890 vertices (SMPL size), 200 000 query points: nearest-vertex ids equal the oracle's brute force, points within 2e-5."""
    from oracle import deform_oracle as do
    from humanliff_amd import synthetic as syn
    from humanliff_amd.NeRF.deform import deform_target2c
    V, P = 6890, 200000
    cpu_model = syn.smpl_like_model(V, 5)
    pose = syn.smpl_like_pose(V, cpu_model, 15, n_points=P)
    model = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in cpu_model.items()}
    can, cd, _, ids = deform_target2c(model, pose, pose["pts"].to(dev), pose["viewdirs"].to(dev), return_ids=True)
    sub = slice(0, 20000)
    ocan, ocd, oid = do.deform_target2c(cpu_model, pose, pose["pts"][0, sub], pose["viewdirs"][0, sub])
    same = ids.cpu()[sub].long() == oid
    assert same.float().mean() > 0.9995        # a near-tie between two vertices may resolve differently in float32
    assert (can.cpu()[0, sub][same] - ocan[same]).abs().max() < 2e-5
    assert (cd.cpu()[0, sub][same] - ocd[same]).abs().max() < 2e-5


def test_canonical_space_render_matches_oracle(dev):
    """use_canonical_space=True end to end: device deformation of every sample point + evaluate-once render, against the oracle's
    deform_target2c + plane lookup + MLP + importance sampling + compositing on the same synthetic body, rays and uniforms."""
    from oracle import deform_oracle as do, render_oracle as orc
    from humanliff_amd import synthetic as syn
    V, H, W, N = 1500, 40, 40, 32
    cpu_model = syn.smpl_like_model(V, 7)
    pose = syn.smpl_like_pose(V, cpu_model, 17, n_points=8)
    mlp = syn.render_mlp_state(3)
    r = make_renderer(mlp, dev)
    r.use_canonical_space = True
    r.SMPL_NEUTRAL = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in cpu_model.items()}
    planes = syn.triplane(seed=11)
    # rays towards the posed body: the orbit camera shifted to the body's world position
    centre = pose["vertices"][0].mean(0)
    lo, hi = pose["vertices"][0].min(0).values - 0.1, pose["vertices"][0].max(0).values + 0.1
    ro, rd, _, _ = syn.orbit_rays(4, 36, H, W)
    ro = ro + centre
    nr, fr = syn.near_far_from_bounds(torch.stack([lo, hi]).double().numpy(), ro.double().numpy(), rd.double().numpy())
    nr, fr = torch.from_numpy(nr).float(), torch.from_numpy(fr).float()
    R = ro.shape[0]
    u = torch.rand((R, N), generator=torch.Generator().manual_seed(9))
    tp = {k: (v if not torch.is_tensor(v) else v) for k, v in pose.items()}
    out = r.render(tp, None, None, ro[None].to(dev), rd[None].to(dev), nr[None].to(dev), fr[None].to(dev), planes.to(dev), N, False,
                   n_samples=N, u=u.to(dev))
    # oracle
    tb = pose["t_world_bounds"][0]
    t = torch.linspace(0.0, 1.0, steps=N)
    z = nr[:, None] * (1.0 - t) + fr[:, None] * t
    vd = rd / rd.norm(dim=1, keepdim=True)

    def evaluate(zz):
        S = zz.shape[1]
        pts = (ro[:, None, :] + rd[:, None, :] * zz[:, :, None]).reshape(-1, 3)
        can, cd, _ = do.deform_target2c(cpu_model, pose, pts, vd[:, None, :].expand(R, S, 3).reshape(-1, 3))
        rgb_raw, sig = ro_mlp(can, cd)
        return rgb_raw.reshape(R, S, 3), sig.reshape(R, S)

    def ro_mlp(can, cd):
        return orc.mlp(mlp, orc.plane_features(planes[0], can, tb), cd)

    with torch.no_grad():
        _, sig_c = evaluate(z)
        z_all = orc.importance_z(sig_c, z, rd, u)
        rgb_raw, sig = evaluate(z_all)
        rgb, acc, depth = orc.composite(rgb_raw, sig, z_all, False)
    got = out["rgb_map"][0].cpu()
    err = (got - rgb).abs().max(dim=1).values
    assert float(acc.max()) > 0.05                       # the body is actually hit
    assert (err < 1e-4).float().mean() > 0.995           # a sample on a Voronoi boundary of the body may pick the other vertex
    assert float(err.median()) < 2e-6
    assert ((out["acc_map"][0].cpu() - acc).abs() < 1e-4).float().mean() > 0.995


@pytest.mark.parametrize("mode", ["random", "local", "bench"])
def test_deform_group_culling_is_the_full_scan(mode, dev, tmp_path):
    """k_deform_rays_cull (nearest vertex among the candidates of a 64-point group) must return what the full scan returns, bit for
    bit - vertex order without locality, index-local order, and the tight-box geometry with silhouette groups that keep every vertex.
    The full scan is selected per process (HL_DEFORM_BRUTE), so both runs go through tests/deform_cull_check.py."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for brute in (False, True):
        env = dict(os.environ)
        env.pop("HL_DEFORM_BRUTE", None)
        if brute:
            env["HL_DEFORM_BRUTE"] = "1"
        out = str(tmp_path / f"{mode}_{int(brute)}.pt")
        args = [sys.executable, os.path.join(root, "tests", "deform_cull_check.py"), out] + ([] if mode == "random" else [mode])
        r = subprocess.run(args, env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(out))
    assert torch.equal(outs[0]["pts"], outs[1]["pts"]) and torch.equal(outs[0]["dirs"], outs[1]["dirs"])
    assert torch.isfinite(outs[0]["pts"]).all()


def test_canonical_density_grid_matches_oracle(dev):
    """extract_geometry's density field with use_canonical_space=True (renderer.py:296-321): lattice over world_bounds, every point through
    deform_target2c, looked up in t_world_bounds."""
    from oracle import deform_oracle as do, render_oracle as orc
    from humanliff_amd import synthetic as syn
    V, N = 1200, 20
    cpu_model = syn.smpl_like_model(V, 7)
    pose = syn.smpl_like_pose(V, cpu_model, 17, n_points=8)
    mlp = syn.render_mlp_state(3)
    r = make_renderer(mlp, dev)
    r.use_canonical_space = True
    r.SMPL_NEUTRAL = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in cpu_model.items()}
    planes = syn.triplane(seed=11, H=64, W=64)
    lo, hi = pose["vertices"][0].min(0).values - 0.1, pose["vertices"][0].max(0).values + 0.1
    tp = dict(pose)
    tp["world_bounds"] = torch.stack([lo, hi])[None]
    u = r.density_grid(tp, planes.to(dev), resolution=N, rays_per_launch=160).cpu()
    X, Y, Z = (torch.linspace(float(lo[k]), float(hi[k]), N) for k in range(3))
    xx, yy, zz = torch.meshgrid(X, Y, Z, indexing="ij")
    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], 1)
    with torch.no_grad():
        can, _, ids = do.deform_target2c(cpu_model, pose, pts, None)
        sig = orc.mlp(mlp, orc.plane_features(planes[0], can, pose["t_world_bounds"][0]))
    err = (u.reshape(-1) + sig).abs()
    assert (err < 5e-5).float().mean() > 0.999          # near-tie nearest vertices may resolve differently in float32
    assert torch.isfinite(u).all()


@pytest.mark.parametrize("seed,burn,shape", [(0, 0, (4096, 128)), (12345, 17, (1000, 32)), (2 ** 40 + 7, 700, (257, 129)), (5, 623, (65536, 1))])
def test_device_mt19937_continues_the_cpu_generator_bit_for_bit(seed, burn, shape, dev):
    """sample_pdf's uniforms (`u = torch.rand(...)` on the CPU generator, NeRF/renderer.py:545) drawn by the device from the CPU generator's
    state: the same bits as torch.rand, from any position of the 624-word block (fresh seed, mid-block, last word), ragged sizes, and the
    CPU generator ends up in exactly the state torch.rand would have left it in - also across chunked consecutive draws."""
    from humanliff_amd.NeRF.cpu_rng import rand_like_cpu
    torch.manual_seed(seed)
    if burn:
        torch.rand(burn)
    st = torch.get_rng_state()
    want = torch.rand(shape)
    want2 = torch.rand(1000)                       # what the NEXT host draw must be
    st_after = torch.get_rng_state()
    torch.set_rng_state(st)
    got, pend = rand_like_cpu(list(shape), dev)
    assert torch.equal(torch.get_rng_state(), st)  # the host generator has not moved yet
    pend.finish()
    assert torch.equal(got.cpu(), want)
    assert torch.equal(torch.rand(1000), want2) and torch.equal(torch.get_rng_state(), st_after)
    # three consecutive device draws = one stream
    torch.set_rng_state(st)
    n = int(np.prod(shape))
    cuts = [0, n // 3, n // 3 + 625, n]
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            p_, pe = rand_like_cpu([b - a], dev)
            pe.finish()
            parts.append(p_)
    assert torch.equal(torch.cat(parts).cpu(), want.reshape(-1))
    # one request walked in pieces (the walk kernel addresses at most 2^28 words per launch; here 4 099 and 624): the same stream, the same final state
    from humanliff_amd import _lib
    for piece in (4099, 624):
        _lib.check(_lib.lib().hl_debug_set_mt19937_piece(piece))
        try:
            torch.set_rng_state(st)
            got, pend = rand_like_cpu(list(shape), dev)
            pend.finish()
            assert torch.equal(got.cpu(), want), piece
            assert torch.equal(torch.rand(1000), want2) and torch.equal(torch.get_rng_state(), st_after), piece
        finally:
            _lib.check(_lib.lib().hl_debug_set_mt19937_piece(0))


def test_render_with_u_none_replays_the_reference_call_on_the_device(dev):
    """The drop-in call (u = None): render() draws the reference's CPU-generator uniforms on the device - same images as with host-drawn
    uniforms, bit for bit, chunked call pattern included, and the CPU generator advanced identically."""
    from humanliff_amd import synthetic as syn
    from humanliff_amd.NeRF import Renderer, render
    rend = Renderer(use_canonical_space=False, triplane_dim=64, triplane_ch=27, smpl_type="smpl", test=True)
    rend.load_state_dict(syn.render_mlp_state(3, gain=2.0), strict=False)
    rend = rend.to(dev)
    g = torch.Generator().manual_seed(11)
    planes = (torch.randn((1, 3, 9, 64, 64), generator=g) * 0.3).clamp(-1, 1).to(dev)
    R = 3000                                        # ragged against the chunk below
    ro = torch.tensor([0.0, 0.0, -3.0]).expand(1, R, 3).contiguous().to(dev)
    rd = torch.nn.functional.normalize(torch.randn((1, R, 3), generator=g) * 0.15 + torch.tensor([0.0, 0.0, 1.0]), dim=-1).to(dev)
    near, far = torch.full((1, R), 2.0, device=dev), torch.full((1, R), 4.0, device=dev)
    tp = {"world_bounds": torch.tensor([[[-1.0, -1.1, -1.0], [1.0, 1.1, 1.0]]], device=dev)}
    outs, states = [], []
    for host in (True, False):
        rend.cpu_uniforms_on_host = host
        torch.manual_seed(77)
        torch.rand(5)
        outs.append(render(chunk=1024, rays_o=ro, rays_d=rd, near=near, far=far, tri_planes=planes, tp_input=tp, renderer=rend, n_samples=32,
                           n_importance=32))
        states.append(torch.get_rng_state())
    assert torch.equal(states[0], states[1])
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert float(outs[0][1].max()) > 0.5            # (the rays hit the volume)
