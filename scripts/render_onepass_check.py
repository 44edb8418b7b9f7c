"""Developer probe: the two-launch one-pass schedule against the four-launch one on a full 512x512 view - bit equality of the images, ms per view of both."""
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
res = {}
for mode in os.environ.get("HL_MODES", "fp16x2,bf16x3").split(","):
    r.mlp_products = mode
    for four in (True, False, True, False):
        r.four_launch = four
        for it in range(23):
            if it == 3:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            out = r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, 128, False, n_samples=128, u=u)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        key = (mode, four)
        img = {k: out[k].clone() for k in ("rgb_map", "acc_map", "depth_map")}
        if key in res:
            assert all(torch.equal(img[k], res[key][k]) for k in img), "not reproducible"
        res[key] = img
        print(f"{mode} {'four-launch' if four else 'one-pass   '}: {ms:.3f} ms per view; rgb mean {float(out['rgb_map'].mean()):.6f}", flush=True)
    a, b = res[(mode, True)], res[(mode, False)]
    for k in a:
        d = (a[k] - b[k]).abs().max()
        print(f"   {mode} {k}: four-launch vs one-pass max-abs {float(d):.3e}  bit-equal {bool(torch.equal(a[k], b[k]))}  finite {bool(torch.isfinite(b[k]).all())}")
