"""Developer probe: device time of the CPU generator's continuation (hl_mt19937_uniform) for one 512x512 view's uniforms, and the drop-in render call (u = None) against resident uniforms."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import synthetic as syn
from humanliff_amd.NeRF import Renderer, cpu_rng
dev = torch.device("cuda:0")
torch.manual_seed(5)
n = 512 * 512 * 128
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    u, pend = cpu_rng.rand_like_cpu((512 * 512, 128), dev)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pend.finish()
    print(f"hl_mt19937_uniform, {n / 1e6:.1f} M numbers: {dt * 1e3:.2f} ms (host call + device)", flush=True)
ref = torch.rand((4, 128))      # the generator keeps going on the host: still the same stream
planes = syn.triplane(seed=11).to(dev)
r = Renderer(use_canonical_space=False, triplane_ch=27, test=True); r.load_state_dict(syn.render_mlp_state(3), strict=False); r = r.to(dev)
ro, rd, nr, fr = [t.to(dev) for t in syn.orbit_rays(3, 36, 512, 512)]
tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
ures = torch.rand((512 * 512, 128), device=dev)
for name, uu in (("resident uniforms", ures), ("u = None (CPU generator continued on the device)", None)):
    for it in range(15):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, 128, False, n_samples=128, u=uu)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 12 * 1e3:.3f} ms per view", flush=True)
