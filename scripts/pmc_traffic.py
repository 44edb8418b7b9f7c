"""Developer probe for the HBM-traffic PMC passes: a calibration copy (known bytes), the dominant conv shape
and one 512x512 render.  Run under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE`."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib, synthetic as syn
from humanliff_amd.NeRF import Renderer
L = _lib.lib(); dev = torch.device("cuda:0")
# calibration: 512 MiB fp32 copy (reads 512 MiB, writes 512 MiB; larger than the 256 MiB Infinity Cache)
a = torch.randn(128 * 1024 * 1024, device=dev); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
# dominant conv: 192->192 3x3 @256x256, B=4, GroupNorm+SiLU prologue
N, C, H, W, Co = 4, 192, 256, 256, 192
x = torch.randn((N, H, W, C), device=dev); w = torch.randn((Co, C, 3, 3), device=dev) * 0.02; bias = torch.randn(Co, device=dev)
cA = torch.rand((N, C), device=dev) + 0.5; cB = torch.randn((N, C), device=dev) * 0.1
res = torch.randn((N, H, W, Co), device=dev)
out = torch.empty((N, H, W, Co), device=dev); scratch = torch.empty(Co * C * 9 + 64, device=dev)
for _ in range(3):
    _lib.check(L.hl_conv2d_nhwc(_lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(bias), Co, 3, 1, 0, _lib.ptr(cA), _lib.ptr(cB), 1, _lib.ptr(res),
                                _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
torch.cuda.synchronize()
# render: one 512x512 view, 128+128
planes = syn.triplane(seed=11).to(dev)
r = Renderer(use_canonical_space=False, triplane_ch=27, test=True); r.load_state_dict(syn.render_mlp_state(3), strict=False); r = r.to(dev)
ro_, rd_, nr_, fr_ = [t.to(dev) for t in syn.orbit_rays(3, 36, 512, 512)]
tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
u = torch.rand((512 * 512, 128), device=dev)
for _ in range(2):
    r.render(tp, None, None, ro_[None], rd_[None], nr_[None], fr_[None], planes, 128, False, n_samples=128, u=u)
torch.cuda.synchronize()
print("done")
