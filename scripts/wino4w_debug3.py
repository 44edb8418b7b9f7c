import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def run(x, w, b, mode):
    N, H, W, C = x.shape; Co = w.shape[0]
    os.environ["HL_WINO4W"] = mode
    out = torch.zeros((N, H, W, Co), device=dev)
    scratch = torch.empty(Co * C * 9 * 5 + 256 + (64 << 20) + N * H * W * C, device=dev)
    _lib.check(L.hl_conv2d_nhwc_mode(0, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, 3, 1, 0, None, None, 0, None, _lib.ptr(out), _lib.ptr(scratch),
                                     scratch.numel() * 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out
N, H, W, Co, C = 4, 128, 128, 192, 16
yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
x = torch.zeros((N, H, W, C), device=dev)
for c in range(C): x[..., c] = (yy * 1000 + xx).float()[None] + c * 0.1
w2 = torch.zeros((Co, C, 3, 3), device=dev)
for o in range(C): w2[o, o, 1, 1] = 1.0
b = torch.zeros(Co, device=dev)
c = run(x, w2, b, "1")
torch.set_printoptions(linewidth=250, precision=1, sci_mode=False)
for ch in (0, 5, 9):
    print(f"channel {ch}: rows 16..23, cols 32..43 (interior block):")
    print(c[0, 16:24, 32:44, ch].cpu())
print("out channels 16..20 at (20,40):", c[0, 20, 40, 14:22].cpu())
