"""The render leg with sample_pdf's uniforms drawn the reference's way (CPU generator, uploaded per view) against device-resident
uniforms: what the drop-in costs when the caller does not set Renderer.uniforms_on_device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import synthetic as syn
from humanliff_amd.NeRF import Renderer

dev = torch.device("cuda:0")
H = W = 512
N = 128
planes = syn.triplane(seed=11).to(dev)
r = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type='smpl', test=True)
r.load_state_dict(syn.render_mlp_state(3), strict=False)
r = r.to(dev)
tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
rays = [[t.to(dev) for t in syn.orbit_rays(v, 36, H, W)] for v in range(6)]
u = torch.rand((H * W, N), device=dev)


def run(mode, views=12):
    r.uniforms_on_device = mode == "device draw"
    kw = {"u": u} if mode == "resident" else {}
    for v in range(2):
        ro, rd, nr, fr = rays[v]
        r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, N, False, n_samples=N, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for v in range(views):
        ro, rd, nr, fr = rays[v % 6]
        r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, N, False, n_samples=N, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / views
    print("UNIFORMS %-12s %.1f ms per 512x512 view = %.2f Mrays/s" % (mode, dt * 1e3, H * W / dt / 1e6))


for m in ("resident", "device draw", "cpu draw"):
    run(m)
t0 = time.perf_counter()
x = torch.rand([H * W, N])
print("UNIFORMS torch.rand on the CPU generator, one view: %.1f ms (%d threads)" % ((time.perf_counter() - t0) * 1e3, torch.get_num_threads()))
