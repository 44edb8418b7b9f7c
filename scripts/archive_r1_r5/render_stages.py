"""Developer probe: one 512x512 view at 128+128 through Renderer.render, default (evaluate-once) and reevaluate=True;
run under `rocprofv3 --kernel-trace --stats` for the per-kernel times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import synthetic as syn
from humanliff_amd.NeRF import Renderer
dev = torch.device("cuda:0")
planes = syn.triplane(seed=11).to(dev)
r = Renderer(use_canonical_space=False, triplane_ch=27, test=True); r.load_state_dict(syn.render_mlp_state(3), strict=False); r = r.to(dev)
ro, rd, nr, fr = [t.to(dev) for t in syn.orbit_rays(3, 36, 512, 512)]
tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
u = torch.rand((512 * 512, 128), device=dev)
for re in (False, True):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, 128, False, n_samples=128, u=u, reevaluate=re)
        torch.cuda.synchronize(); print("reevaluate", re, f"{(time.perf_counter() - t0) * 1e3:.2f} ms")
