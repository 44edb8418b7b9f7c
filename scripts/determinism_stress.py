"""Stress probe: repeat the production UNet forward (batch 4 and 1) and one 512x512 render many times and require bit-identical
outputs every time - any missed wait / race in the DMA pipelines shows up as run-to-run differences."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from humanliff_amd import synthetic as syn
from humanliff_amd.NeRF import Renderer
dev = torch.device("cuda:0")
model, diffusion, sd = bench.build_unet(dev)
g = torch.Generator().manual_seed(3)
bad = 0
for B, reps in ((4, 40), (1, 40), (3, 20)):
    x = torch.randn((B, 27, 256, 256), generator=g).to(dev); xc = torch.randn((B, 27, 256, 256), generator=g).to(dev) * 0.3
    t = torch.randint(0, 1000, (B,), generator=g).to(dev); y = torch.randint(0, 5, (B,), generator=g).to(dev)
    with torch.no_grad():
        ref = model(x, t, xc, y=y).clone()
        for i in range(reps):
            out = model(x, t, xc, y=y)
            if not torch.equal(out, ref):
                bad += 1; print("UNet mismatch B", B, "rep", i, float((out - ref).abs().max()))
    print("UNet B", B, "reps", reps, "ok" if bad == 0 else "MISMATCHES", flush=True)
planes = syn.triplane(seed=11).to(dev)
r = Renderer(use_canonical_space=False, triplane_ch=27, test=True); r.load_state_dict(syn.render_mlp_state(3), strict=False); r = r.to(dev)
ro, rd, nr, fr = [t.to(dev) for t in syn.orbit_rays(3, 36, 512, 512)]
tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
u = torch.rand((512 * 512, 128), device=dev)
ref = {k: v.clone() for k, v in r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, 128, False, n_samples=128, u=u).items()}
for i in range(15):
    out = r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, 128, False, n_samples=128, u=u)
    for k in ("rgb_map", "acc_map", "depth_map"):
        if not torch.equal(out[k], ref[k]):
            bad += 1; print("render mismatch", k, "rep", i)
print("render reps 15", "ok" if bad == 0 else "MISMATCHES")
print("TOTAL MISMATCHES", bad)
