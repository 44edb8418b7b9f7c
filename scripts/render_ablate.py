"""Developer probe: time the fine/coarse ray-march kernels with ablation builds (scripts/rexp_*.so)."""
import sys, os, glob, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib, synthetic as syn
from humanliff_amd.NeRF import Renderer
dev = torch.device("cuda:0")
planes = syn.triplane(seed=11).to(dev)
r = Renderer(use_canonical_space=False, triplane_ch=27, test=True); r.load_state_dict(syn.render_mlp_state(3), strict=False); r = r.to(dev)
packed, pp = r._packed_mlp(dev), r._packed_planes(planes[0])
ro, rd, nr, fr = [t.to(dev).contiguous() for t in syn.orbit_rays(3, 36, 512, 512)]
bd = torch.tensor(syn.WORLD_BOUNDS).to(dev); R = 512 * 512; N = 128
sig = torch.empty(R * N, device=dev); z_all = torch.empty(R * 2 * N, device=dev)
u = torch.rand((R, N), device=dev)
rgb, acc, dep = torch.empty((R, 3), device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev)
p, s = _lib.ptr, _lib.stream_ptr
L0 = _lib.lib()
L0.hl_render_coarse(p(packed), p(pp), 256, 256, p(bd), p(ro), p(rd), p(nr), p(fr), None, R, N, p(sig), s())
L0.hl_render_importance(p(sig), p(rd), p(nr), p(fr), None, p(u), R, N, N, p(z_all), s())
torch.cuda.synchronize()
for rnd in range(2):
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "rexp_*.so"))):
        L = C.CDLL(path)
        fine = L.hl_render_fine; fine.restype = C.c_int; fine.argtypes = _lib.SIGNATURES["hl_render_fine"][1]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        fine(p(packed), p(pp), 256, 256, p(bd), p(ro), p(rd), p(nr), p(fr), p(z_all), 1, R, 2 * N, 2, p(rgb), p(acc), p(dep), s())
        ev[0].record()
        fine(p(packed), p(pp), 256, 256, p(bd), p(ro), p(rd), p(nr), p(fr), p(z_all), 1, R, 2 * N, 2, p(rgb), p(acc), p(dep), s())
        ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1])
        print(f"{os.path.basename(path):44s} fine {ms:7.2f} ms {R*256*132608/ms/1e9:6.1f} TF/s", flush=True)
