"""Developer probe: time per 512x512 view of the bf16x3 renderer under HL_B3_ABL (set in the environment before the library loads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import synthetic as syn
from humanliff_amd.NeRF import Renderer, render_view
dev = torch.device("cuda:0")
rend = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type="smpl", test=True)
rend.load_state_dict(syn.render_mlp_state(3), strict=False)
rend = rend.to(dev)
rend.mlp_products = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
planes = (0.3 * torch.randn((1, 3, 9, 256, 256), generator=torch.Generator().manual_seed(11))).clamp(-1, 1).to(dev)
tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
res = 512
u = torch.rand((res * res, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
def cam(v):
    K, c2w, c = syn.orbit_camera(v, 36, res, res)
    R = c2w.T.copy()
    return K, R, (-R @ c).reshape(3, 1)
render_view(res, res, *cam(0), planes, tp, rend, n_samples=128, n_importance=128, u=u)
torch.cuda.synchronize(); t0 = time.perf_counter()
for v in (0, 7, 19, 23):
    render_view(res, res, *cam(v), planes, tp, rend, n_samples=128, n_importance=128, u=u)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
print(f"HL_B3_ABL={os.environ.get('HL_B3_ABL', '0'):>2s} {rend.mlp_products}: {dt * 1e3:.1f} ms per view = {res * res / dt / 1e6:.2f} Mrays/s", flush=True)
