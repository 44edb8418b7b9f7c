import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def run(x, w, b, mode, gn=None):
    N, H, W, C = x.shape; Co = w.shape[0]
    os.environ["HL_WINO4W"] = mode
    out = torch.zeros((N, H, W, Co), device=dev)
    scratch = torch.empty(Co * C * 9 * 5 + 256 + (64 << 20) + N * H * W * C, device=dev)
    _lib.check(L.hl_conv2d_nhwc_mode(0, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, 3, 1, 0, None, None, 0, None, _lib.ptr(out), _lib.ptr(scratch),
                                     scratch.numel() * 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out
for C in (16, 32, 64):
    N, H, W, Co = 4, 128, 128, 192
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, W, C), device=dev, generator=g); w = torch.randn((Co, C, 3, 3), device=dev, generator=g) * 0.1; b = torch.zeros(Co, device=dev)
    a, c = run(x, w, b, "0"), run(x, w, b, "1")
    d = (a - c).abs()
    print(f"C={C}: max diff {float(d.max()):.3e}; per-channel-block max:", [f"{float(d[..., k*32:(k+1)*32].max()):.2e}" for k in range(Co // 32)])
    print("  per image:", [f"{float(d[n].max()):.2e}" for n in range(N)])
    blk = d[0].reshape(H // 16, 16, W // 32, 32, Co).amax(dim=(1, 3, 4))
    print("  per 32x16 block (image 0) max:", [f"{float(v):.1e}" for v in blk.flatten()[:8]])
    t = d[0, :16, :32].reshape(4, 4, 8, 4, Co).amax(dim=(1, 3, 4))
    print("  per tile of block 0:", [[f"{float(v):.1e}" for v in r] for r in t])
    pix = d[0, :4, :4].amax(dim=2)
    print("  pixels of tile 0:", [[f"{float(v):.1e}" for v in r] for r in pix])
    # which input channel is at fault: one-hot channel inputs
    for ci in range(0, C, max(1, C // 8)):
        x1 = torch.zeros_like(x); x1[..., ci] = x[..., ci]
        dd = float((run(x1, w, b, "0") - run(x1, w, b, "1")).abs().max())
        print(f"    only input channel {ci}: diff {dd:.2e}")
    # which frequency: constant image (only frequency (0,0)-ish content) 
    x2 = torch.ones_like(x)
    print("  constant image diff:", float((run(x2, w, b, "0") - run(x2, w, b, "1")).abs().max()))
