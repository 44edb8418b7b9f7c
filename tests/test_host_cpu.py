"""Host-side (no GPU) checks of the product package: state_dict layout, schedules, factories,
the C-ABI library (loads, exports every declared symbol), loud failure without a GPU."""
import ast
import os
import re

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(over):
    from humanliff_amd.improved_diffusion.script_util import model_and_diffusion_defaults
    a = model_and_diffusion_defaults()
    a.update(dict(in_channels=27, out_channels=27, class_cond=True, learn_sigma=False, num_heads=4, rescale_timesteps=False))
    a.update(over)
    return a


@pytest.mark.parametrize("name", ["tiny32", "mid64", "deep256"])
def test_unet_state_dict_layout_equals_reference(name):
    """Keys, order and shapes of UNetModel.state_dict() == the reference's (stored in the fixture)."""
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    over = dict(image_size=int(g["arg_image_size"]), num_channels=int(g["arg_num_channels"]),
                num_res_blocks=int(g["arg_num_res_blocks"]), attention_resolutions=str(g["arg_attention_resolutions"]))
    model, diffusion = create_model_and_diffusion(**_args(over))
    mine = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    ref = [(str(k), ast.literal_eval(str(s))) for k, s in zip(g["keys"], g["shapes"])]
    assert mine == ref
    assert sum(p.numel() for p in model.parameters()) == int(g["n_params"])
    # zero_module'd tensors start at zero like the reference (unet.py:170, 240, 470, 486)
    sd = model.state_dict()
    for k in sd:
        if re.search(r"out_layers\.3\.|proj_out\.|^out\.2\.|input_blocks_proj_cond", k):
            assert float(sd[k].abs().max()) == 0.0, k


def test_production_config_counts():
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion
    m, d = create_model_and_diffusion(**_args(dict(image_size=256, num_channels=192, num_res_blocks=3,
                                                    attention_resolutions="32,16,8", timestep_respacing="250")))
    assert sum(p.numel() for p in m.parameters()) == 497173083
    assert len(m.state_dict()) == 953
    assert d.num_timesteps == 250 and d.timestep_map[:5] == [0, 4, 8, 12, 16] and d.timestep_map[-1] == 999


@pytest.mark.parametrize("tag,spec", [("full", ""), ("r250", "250"), ("ddim50", "ddim50"), ("ddim10", "ddim10"),
                                      ("mix", "10,15,20")])
def test_spaced_diffusion_tables_equal_reference(tag, spec):
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    g = np.load(os.path.join(GOLDEN, "diffusion_steps.npz"))
    d = create_gaussian_diffusion(steps=1000, timestep_respacing=spec)
    assert d.timestep_map == list(g[f"{tag}_map"])
    assert np.array_equal(d.betas, g[f"{tag}_betas"])
    assert np.array_equal(d.posterior_log_variance_clipped, g[f"{tag}_post_logvar"])
    assert np.array_equal(d.posterior_mean_coef1, g[f"{tag}_coef1"])
    assert np.array_equal(d.posterior_mean_coef2, g[f"{tag}_coef2"])
    assert np.array_equal(d.sqrt_recip_alphas_cumprod, g[f"{tag}_sqrt_recip"])
    assert np.array_equal(d.sqrt_recipm1_alphas_cumprod, g[f"{tag}_sqrt_recipm1"])


def test_schedule_errors_match_reference_types():
    from humanliff_amd.improved_diffusion import gaussian_diffusion as gd
    from humanliff_amd.improved_diffusion.respace import space_timesteps
    from humanliff_amd.improved_diffusion.script_util import create_model
    g = np.load(os.path.join(GOLDEN, "diffusion_steps.npz"))
    assert np.array_equal(gd.get_named_beta_schedule("cosine", 50), g["cosine50_betas"])
    with pytest.raises(NotImplementedError):
        gd.get_named_beta_schedule("quadratic", 10)
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")
    with pytest.raises(ValueError):
        space_timesteps(10, "20")
    with pytest.raises(ValueError):       # script_util.py:124
        create_model(100, 27, 32, 27, 1, False, True, False, "16,8", 4, -1, True, "controlnet", False, 0.0)


def test_library_exports_every_declared_symbol():
    from humanliff_amd import _lib
    from humanliff_amd.build import build
    build()
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "humanliff_hip.h")).read()
    declared = set(re.findall(r"\b(hl_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"hl_status"}
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/humanliff_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert L.hl_version() >= 100
    assert L.hl_render_mlp_packed_bytes() == (17 * 4096 + 1024) * 4 + 132 * 1024 + 3 * 132 * 1024 + 2 * 132 * 1024      # fp32 image + the 132 fp16 fragments of k_march16 + the three bf16 planes (bf16x3) + the two fp16 planes (fp16x2)
    assert L.hl_planes_packed_bytes(256, 256) == 9 * 256 * 256 * 16


def test_no_cpu_fallback():
    """CPU tensors must fail loudly, never silently route to PyTorch."""
    from humanliff_amd.NeRF import Renderer
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion
    m, d = create_model_and_diffusion(**_args(dict(image_size=32, num_channels=32, num_res_blocks=1)))
    x = torch.zeros(1, 27, 32, 32)
    with pytest.raises(RuntimeError):
        m(x, torch.tensor([3]), x, y=torch.tensor([0]))
    with pytest.raises(RuntimeError):
        d.p_sample(lambda *a, **k: x, x, x, torch.tensor([3]))
    r = Renderer(use_canonical_space=False, triplane_ch=27, test=True)
    with pytest.raises((RuntimeError, AssertionError)):
        r.render({"world_bounds": torch.zeros(1, 2, 3)}, None, None, torch.zeros(1, 4, 3), torch.ones(1, 4, 3),
                 torch.zeros(1, 4), torch.ones(1, 4), torch.zeros(1, 3, 9, 8, 8), 0, False, n_samples=4)


def test_product_never_imports_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "humanliff_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert "oracle" not in src, f"{f} mentions the oracle"


def test_recon_twin_mirrors_reference_names():
    """recon_NeRF/lib/renderer.py:13-29, 244 and run_nerf_batch.py:29: constructor / render argument names and the tri_planes Parameter."""
    import inspect
    from humanliff_amd.recon_NeRF import Renderer, render
    r = Renderer(use_canonical_space=False, num_instances=2, triplane_dim=16, triplane_ch=27, test=False)
    assert list(inspect.signature(Renderer.__init__).parameters)[1:] == ["use_canonical_space", "num_instances", "triplane_dim", "triplane_ch", "test"]
    assert tuple(r.tri_planes.shape) == (2, 4, 3, 9, 16, 16) and r.tri_planes.requires_grad
    assert list(inspect.signature(r.render).parameters)[:9] == ["tp_input", "world_pts", "z_vals", "rays_o", "rays_d", "near", "far",
                                                                 "n_importance", "white_bkgd"]
    assert list(inspect.signature(render).parameters) == ["chunk", "rays_o", "rays_d", "near", "far", "tp_input", "renderer", "n_samples",
                                                          "perturb", "n_importance", "white_bkgd"]
    keys = set(r.state_dict().keys())
    assert {"tri_planes", "pts_linears.0.weight", "alpha_linear.bias", "views_linear.weight", "rgb_linear.bias"} <= keys


def test_spaced_diffusion_wrapper_keeps_temporary_callables_alive_and_releases_models():
    """SpacedDiffusion._wrap_model: the wrapper handed to a loop holds its model strongly (a lambda / functools.partial / bound method passed
    inline must survive the loop: round-3 advisor finding), the per-model cache holds it weakly (deleting a model frees it) and the device
    tensor of the timestep map is built once per model and (device, dtype), not per call like the reference (respace.py:117-122)."""
    import functools
    import gc
    import weakref
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="ddim10")

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x, t, x_cond, **kw):
            return x * 0 + t.float().reshape(-1, 1)

    m = M()
    for make in (lambda: (lambda x, t, xc, **kw: m(x, t, xc, **kw)), lambda: functools.partial(m.forward), lambda: m.forward):
        w = d._wrap_model(make())                 # the only other reference to the callable dies here
        gc.collect()
        out = w(torch.zeros(1, 2), torch.tensor([3]), None)
        assert float(out[0, 0]) == 300.0          # step 3 of ddim10 = original timestep 300
    w1, w2 = d._wrap_model(m), d._wrap_model(m)
    w1(torch.zeros(1, 2), torch.tensor([1]), None)
    assert w1._maps is w2._maps and len(w1._maps) == 1
    r = weakref.ref(m)
    del m, w, w1, w2
    gc.collect()
    assert r() is None and len(d._wrapped) == 0


def test_extract_into_tensor_tables_are_keyed_by_content():
    """_extract_into_tensor (gaussian_diffusion.py:850-863 of the reference: from_numpy(arr).to(device)[t].float(), broadcast) keeps one device copy
    per table CONTENT: callers pass temporaries (1.0 - alphas_cumprod) whose addresses the allocator reuses - a second array at the same
    address with other values must not hit the first one's copy - and the values are the reference expression's."""
    import torch
    from humanliff_amd.improved_diffusion import gaussian_diffusion as gd
    t = torch.tensor([0, 3, 999, 500])
    rng = np.random.default_rng(0)
    for _ in range(3):
        arr = rng.random(1000)                      # fp64, like the schedule tables
        got = gd._extract_into_tensor(arr, t, (4, 2, 3))
        ref = torch.from_numpy(arr)[t].float()
        assert got.shape == (4, 2, 3) and torch.equal(got[:, 0, 0], ref) and torch.equal(got[:, 1, 2], ref)
        arr[...] = rng.random(1000)                 # same address, new content
        assert torch.equal(gd._extract_into_tensor(arr, t, (4,)), torch.from_numpy(arr)[t].float())
    arr32 = rng.random(1000).astype(np.float32)     # another dtype with other bytes of the same length is its own entry
    assert torch.equal(gd._extract_into_tensor(arr32, t, (4,)), torch.from_numpy(arr32)[t].float())
    assert len(gd._TABLES) <= 256


def test_bench_line_is_compact():
    """the driver reads ONE JSON line from bench.py's stdout.  Round 5's line had grown to 20 KB and the driver's record came back
    unparsed; `bench_line.compact_line` now reduces everything measured to < 6 KB with no string over 160 characters (the rest goes to
    bench_detail.json).  Canned input: round 5's full result (profiles/r05_bench.json) plus a worst-case synthetic one."""
    import json
    import os
    import bench_line
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, "profiles", "r05_bench.json")))
    line = bench_line.compact_line(full)
    s = json.dumps(line)
    assert len(s) < 8192 and len(s) < bench_line.MAX_LINE_BYTES and "\n" not in s
    assert max(len(x) for x in bench_line._strings(line)) <= 160
    back = json.loads(s)
    assert back == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline", "parity", "summary"):
        assert k in back, k
    assert back["value"] == full["value"] and back["metric"] == "denoise-steps/sec" and "workload" in back["config"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"], k
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    for leg in ("render", "fit"):
        assert set(("value", "unit", "roofline", "cpu_baseline")) <= set(back[leg]), leg
    assert back["render"]["parity"]["psnr_db"] > 45
    # worst case: every string absurdly long, non-finite numbers, an N > 1 run without the N = 1-only legs
    junk = "x" * 5000
    worst = dict(full, dtype_note=junk, cpu_baseline=None, parity=None, e2e=None, train=None,
                 rccl={"world_size": 8, "backend": "nccl", "device_count_visible": 8, "sample_gather": {"recv_gb_per_s_per_rank": float("nan")},
                       "image_gather_uint8": {"recv_gb_per_s_per_rank": 301.5, "note": junk}})
    worst["config"] = dict(full["config"], workload=junk)
    worst["roofline"] = dict(full["roofline"], kernel=junk, note=junk)
    w = bench_line.compact_line(worst)
    ws = json.dumps(w)
    assert len(ws) < bench_line.MAX_LINE_BYTES and json.loads(ws) == w and "NaN" not in ws
    assert w["rccl"]["backend"] == "nccl" and w["rccl"]["image_gather_uint8_recv_gb_per_s_per_rank"] == 301.5
