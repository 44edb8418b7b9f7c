import os, sys
sys.path.insert(0, "/root/repo")
import torch
from humanliff_amd import _lib, synthetic as syn
from humanliff_amd.NeRF import Renderer
dev = torch.device("cuda:0"); L = _lib.lib()
H = W = 512; N = 128
planes = syn.triplane(seed=11).to(dev)
r = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, smpl_type='smpl', test=True)
r.load_state_dict(syn.render_mlp_state(3), strict=False); r = r.to(dev)
tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}
ro, rd, nr, fr = [t.to(dev).contiguous() for t in syn.orbit_rays(0, 36, H, W)]
u = torch.rand((H * W, N), device=dev)
for mode in (False, True):
    r.mlp_fp16 = mode
    for _ in range(2): r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, N, False, n_samples=N, u=u)
torch.cuda.synchronize()
