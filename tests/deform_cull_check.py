"""Developer check: hl_deform_rays output (canonical points / directions of every sample) -> file, for comparing the group-culling
kernel against the full scan (HL_DEFORM_BRUTE=1) bit for bit.   python tests/deform_cull_check.py out.pt [local]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from humanliff_amd import _lib, synthetic as syn
from humanliff_amd.NeRF.deform import deform_tables

dev = torch.device("cuda:0")
model = syn.smpl_like_model()
if len(sys.argv) > 2 and sys.argv[2] == 'local':     # index-local vertex order (like a real mesh): sort the template along a space-filling-ish key
    v = model["v_template"]
    key = (v[:, 1] * 8).floor() * 64 + (v[:, 0] * 8).floor() * 8 + (v[:, 2] * 8).floor()
    order = torch.argsort(key)
    for k in ("v_template", "shapedirs", "posedirs", "weights"):
        model[k] = model[k][order]
    model["J_regressor"] = model["J_regressor"][:, order]
pose = syn.smpl_like_pose(6890, model, seed=31, n_points=16)
md = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in model.items()}
todev = lambda d: {k: v.to(dev) for k, v in d.items()}
verts4, table, Rh, Th = deform_tables(md, todev(pose["params"]), todev(pose["t_params"]), pose["vertices"].to(dev))
H = W = 256
ro, rd, nr, fr = [t.to(dev).contiguous() for t in syn.orbit_rays(3, 8, H, W)]
# put the rays through the posed body's box
c = pose["vertices"][0].mean(0).to(dev)
ro = (ro + c).contiguous()
if len(sys.argv) > 2 and sys.argv[2] == "bench":      # the geometry of scripts/canonical_bench.py: tight box around the body, 512x512
    H = W = 512
    lo, hi = pose["vertices"][0].min(0).values - 0.1, pose["vertices"][0].max(0).values + 0.1
    ro, rd, _, _ = syn.orbit_rays(4, 36, H, W)
    ro = ro + pose["vertices"][0].mean(0)
    nr, fr = syn.near_far_from_bounds(torch.stack([lo, hi]).double().numpy(), ro.double().numpy(), rd.double().numpy())
    ro, rd, nr, fr = ro.to(dev).contiguous(), rd.to(dev).contiguous(), torch.from_numpy(nr).float().to(dev), torch.from_numpy(fr).float().to(dev)
R, S = H * W, 128
T32 = (R + 31) // 32 * 32
pts = torch.empty((T32 * S, 4), device=dev); dirs = torch.empty((T32 * S, 4), device=dev); scr = torch.zeros(4, device=dev)
L = _lib.lib(); p = _lib.ptr
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    _lib.check(L.hl_deform_rays(p(ro), p(rd), p(nr), p(fr), None, 0, R, S, Rh.ctypes.data, Th.ctypes.data, p(verts4), p(table), int(verts4.shape[0]),
                                p(pts), p(dirs), p(scr), _lib.stream_ptr()), "hl_deform_rays")
    torch.cuda.synchronize(); dt = time.time() - t0
print(f"hl_deform_rays {R} rays x {S}: {dt * 1e3:.2f} ms")
torch.save({"pts": pts.cpu(), "dirs": dirs[:, :3].cpu()}, sys.argv[1])
