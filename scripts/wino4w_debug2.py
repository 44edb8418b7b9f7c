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
N, H, W, Co = 4, 128, 128, 192
for C in (16, 32):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, W, C), device=dev, generator=g); w = torch.randn((Co, C, 3, 3), device=dev, generator=g) * 0.1; b = torch.zeros(Co, device=dev)
    a, c1, c2 = run(x, w, b, "0"), run(x, w, b, "1"), run(x, w, b, "1")
    print(f"C={C}: deterministic {bool(torch.equal(c1, c2))}; corr(a, c) = {float(torch.corrcoef(torch.stack([a.flatten(), c1.flatten()]))[0,1]):.4f}; |a| {float(a.abs().mean()):.3f} |c| {float(c1.abs().mean()):.3f}")
    # delta weights: only centre tap, identity on channels -> output channel o = input channel o (o < C)
    w2 = torch.zeros_like(w)
    for o in range(C): w2[o, o, 1, 1] = 1.0
    a2, c2_ = run(x, w2, b, "0"), run(x, w2, b, "1")
    print("  identity weights: ref err", float((a2[..., :C] - x).abs().max()), "w4w err", float((c2_[..., :C] - x).abs().max()))
    d = (c2_[..., :C] - x)[0]
    print("  w4w - x, image 0, pixel rows 0..5, cols 0..7, channel 0:\n", d[:6, :8, 0].cpu().numpy().round(2))
    print("  w4w out vs x at (0,0..7,ch0):", c2_[0, 0, :8, 0].cpu().numpy().round(2), x[0, 0, :8, 0].cpu().numpy().round(2))
    # is the w4w output equal to x shifted / other channel?
    for ch in range(min(C, 8)):
        e = (c2_[0, :, :, 0] - x[0, :, :, ch]).abs().max()
        print(f"    out ch0 vs in ch{ch}: {float(e):.3f}", end=";")
    print()
