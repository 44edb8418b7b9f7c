"""Developer probe: max-abs errors of the renderer against the reference's golden renders (tests/golden/render_*.npz), every case x product mode."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import load_render_case
from tests.test_render_gpu import hip_render
dev = torch.device("cuda:0")
for name in "abcdef":
    i, e = load_render_case(name)
    for products in ("fp16x2", "bf16x3", "fp32"):
        r, out = hip_render(i, dev, products=products)
        R, N = i["rays_o"].shape[0], i["n_samples"]
        tiles = (R + 31) // 32
        rec = r._ws.cpu()[:tiles * N * 32 * 4].reshape(tiles, N, 32, 4)
        sigma = rec[..., 0].permute(0, 2, 1).reshape(tiles * 32, N)[:R]
        print(f"{name} {products:7s}: sigma {float((sigma - e['sigma_coarse']).abs().max()):.2e} (|sigma| max {float(e['sigma_coarse'].abs().max()):.1f}) "
              f"rgb {float((out['rgb_map'] - e['rgb']).abs().max()):.2e} acc {float((out['acc_map'] - e['acc']).abs().max()):.2e} depth {float((out['depth_map'] - e['depth']).abs().max()):.2e}", flush=True)
