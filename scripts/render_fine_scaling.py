"""Developer probe: hl_render_fine time vs number of samples (fixed per-launch cost vs per-sample cost)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib, synthetic as syn
from humanliff_amd.NeRF import Renderer
L = _lib.lib(); dev = torch.device("cuda:0")
planes = syn.triplane(seed=11).to(dev)
r = Renderer(use_canonical_space=False, triplane_ch=27, test=True); r.load_state_dict(syn.render_mlp_state(3), strict=False); r = r.to(dev)
ro, rd, nr, fr = [t.to(dev).contiguous() for t in syn.orbit_rays(3, 36, 512, 512)]
bd = torch.tensor(syn.WORLD_BOUNDS).to(dev).contiguous()
packed, pp = r._packed_mlp(dev), r._packed_planes(planes[0])
R = 512 * 512
rgb, acc, dep = torch.empty((R, 3), device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev)
p, s = _lib.ptr, _lib.stream_ptr
for S in (32, 64, 128, 256):
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.hl_render_fine(p(packed), p(pp), 256, 256, p(bd), p(ro), p(rd), p(nr), p(fr), None, 0, R, S, 2, p(rgb), p(acc), p(dep), s()))
        e1.record(); torch.cuda.synchronize()
    print(f"S={S}: {e0.elapsed_time(e1):.2f} ms")
