import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
for (N, H, W, C, Co, ks) in ((2, 64, 64, 96, 192, 3), (2, 256, 256, 192, 192, 3), (2, 256, 256, 384, 192, 1)):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, W, C), device=dev, generator=g); w = torch.randn((Co, C, ks, ks), device=dev, generator=g) / (C * ks * ks) ** 0.5; b = torch.zeros(Co, device=dev)
    scratch = torch.empty(Co * C * ks * ks * 5 + 256 + (64 << 20), device=dev)
    outs = {}
    for mode in (2, 0, 1, 4):
        out = torch.zeros((N, H, W, Co), device=dev)
        def call():
            _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, ks, 1, 0, None, None, 0, None, _lib.ptr(out), _lib.ptr(scratch),
                                             scratch.numel() * 4, _lib.stream_ptr()))
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        outs[mode] = (out.clone(), e0.elapsed_time(e1) / 10 * 1e3)
    ref = outs[2][0].double()
    print(f"N{N} {H}x{W} {C}->{Co} {ks}x{ks}: " + "; ".join(f"mode {m}: {t:.0f} us, rel-L2 vs direct fp32 {float((o.double() - ref).norm() / ref.norm()):.2e}" for m, (o, t) in outs.items()))
