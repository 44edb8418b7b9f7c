"""Developer check: activation matrix of hl_render_eval_acts against the oracle's layer outputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tests.test_oracle_render_grad import load_grad_case
from tests.test_render_train_gpu import make_renderer
from humanliff_amd import _lib
from humanliff_amd.NeRF.renderer import untile_rows
from oracle import render_oracle as ro

dev = torch.device("cuda:0")
i, g = load_grad_case("a")
r = make_renderer(i["mlp"], dev)
L = _lib.lib()
R, N = i["z"].shape
T32 = (R + 31) // 32 * 32
P = T32 * N
planes = i["planes"][0].to(dev)
packed, pp = r._packed_mlp(dev), r._packed_planes(planes)
act = torch.full((630, P), float("nan"), device=dev)
vc = torch.empty(T32 * N * 4, device=dev)
f = lambda t: t.to(dev).contiguous()
ro_, rd_, nr_, fr_, bd_, z_ = f(i["rays_o"]), f(i["rays_d"]), f(i["near"]), f(i["far"]), f(i["bounds"]), f(i["z"])
p = _lib.ptr
_lib.check(L.hl_render_eval_acts(p(packed), p(pp), 32, 32, p(bd_), p(ro_), p(rd_), p(nr_), p(fr_), p(z_), 0, R, N, p(vc), p(act), P, 0, _lib.stream_ptr()), "x")
torch.cuda.synchronize()
act = act.cpu()
print("nan count per row block:", [int(torch.isnan(act[a:b]).sum()) for a, b in ((0, 27), (27, 155), (155, 283), (283, 411), (411, 539), (539, 566), (566, 630))])
# oracle
pts = (i["rays_o"][:, None] + i["rays_d"][:, None] * i["z"][:, :, None]).reshape(-1, 3)
feats = ro.plane_features(i["planes"][0], pts, i["bounds"])
m = i["mlp"]
x0 = F.softplus(F.linear(feats, m["pts_linears.0.weight"], m["pts_linears.0.bias"]))
x1 = F.softplus(F.linear(x0, m["pts_linears.1.weight"], m["pts_linears.1.bias"]))
x2 = F.softplus(F.linear(torch.cat([feats, x1], 1), m["pts_linears.2.weight"], m["pts_linears.2.bias"]))
def rows(a, b):   # (b-a, P) tile-major -> (R*N, b-a)
    return torch.stack([untile_rows(act[k], R, N).reshape(-1) for k in range(a, b)], 1)
for nm, (a, b), ref in (("f", (0, 27), feats), ("x1", (27, 155), x1), ("x0", (155, 283), x0), ("x2", (283, 411), x2)):
    got = rows(a, b)
    print(nm, float((got - ref).abs().max()), float(ref.abs().max()))
    if nm == "f":
        print(" per feature err", [(round(float((got[:, k] - ref[:, k]).abs().max()), 5)) for k in range(27)])
