"""Developer probe (GPU box): print per-stage errors of the HIP renderer vs oracle + a timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import load_render_case, psnr
from tests.test_render_gpu import hip_render
from humanliff_amd import synthetic as syn

dev = torch.device("cuda:0")
for name in ["a", "b", "c"]:
    i, e = load_render_case(name)
    r, out = hip_render(i, dev)
    R, N = i["rays_o"].shape[0], i["n_samples"]
    tiles = (R + 31) // 32   # workspace: raw (sigma, r, g, b) records of the coarse points first, tile-major
    sigma = r._ws.cpu()[:tiles * N * 32 * 4].reshape(tiles, N, 32, 4)[..., 0].permute(0, 2, 1).reshape(tiles * 32, N)[:R]
    print(name, "sigma", float((sigma - e["sigma_coarse"]).abs().max()),
          "rgb", float((out["rgb_map"] - e["rgb"]).abs().max()),
          "acc", float((out["acc_map"] - e["acc"]).abs().max()),
          "depth", float((out["depth_map"] - e["depth"]).abs().max()), "psnr", psnr(out["rgb_map"], e["rgb"]), flush=True)

from humanliff_amd.NeRF import Renderer, render
planes = syn.triplane(seed=11).to(dev)
r = Renderer(use_canonical_space=False, triplane_ch=27, test=True)
r.load_state_dict(syn.render_mlp_state(3), strict=False)
r = r.to(dev)
ro_, rd_, nr_, fr_ = [t.to(dev) for t in syn.orbit_rays(3, 36, 512, 512)]
tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
u = torch.rand((512 * 512, 128), device=dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out = r.render(tp, None, None, ro_[None], rd_[None], nr_[None], fr_[None], planes, 128, False, n_samples=128, u=u)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("512x512 view: %.1f ms -> %.3f Mrays/s" % (dt * 1e3, 512 * 512 / dt / 1e6), flush=True)
print("rgb mean", float(out["rgb_map"].mean()), "acc mean", float(out["acc_map"].mean()))
