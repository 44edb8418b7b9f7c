"""Developer probe: one 512x512 view at 128 + 128 in each product mode: ms per view and rgb max-abs / PSNR against the fp32-MFMA kernel."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import synthetic as syn
from humanliff_amd.NeRF import Renderer
dev = torch.device("cuda:0")
H = W = 512; N = 128
planes = syn.triplane(seed=11).to(dev)
r = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type='smpl', test=True)
r.load_state_dict(syn.render_mlp_state(3), strict=False); r = r.to(dev)
tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
rays = [[t.to(dev) for t in syn.orbit_rays(v, 36, H, W)] for v in range(4)]
u = torch.rand((H * W, N), generator=torch.Generator(device=dev).manual_seed(5), device=dev)
def one(v):
    ro, rd, nr, fr = rays[v]
    return r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, N, False, n_samples=N, u=u)
ref = None
for mode in ("fp32", "bf16x3", "fp16x2"):
    r.mlp_products = mode
    one(3); torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [one(v) for v in range(3)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    if ref is None: ref = outs
    err = max(float((o["rgb_map"] - q["rgb_map"]).abs().max()) for o, q in zip(outs, ref))
    mse = max(float(((o["rgb_map"] - q["rgb_map"]) ** 2).mean()) for o, q in zip(outs, ref))
    print(f"{mode}: {dt * 1e3:.2f} ms per view = {H * W / dt / 1e6:.2f} Mrays/s; rgb max-abs vs fp32 kernel {err:.2e}, PSNR {(-10 * math.log10(mse)) if mse else 200:.1f} dB", flush=True)
