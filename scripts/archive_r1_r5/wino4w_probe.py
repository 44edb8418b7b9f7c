"""Developer probe: k_conv_wino4w (64 output channels per workgroup, one wave per SIMD) against k_conv_wino4 on the layer shapes of the
production UNet at batch 4: outputs compared (same arithmetic, so the difference should be zero) and both timed with events.
HL_WINO4W=0/1 is read by conv2d on every call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
shapes = [(4, 256, 256, 192, 192, 0), (4, 128, 128, 192, 192, 0), (4, 64, 64, 384, 384, 0), (4, 64, 64, 768, 384, 0), (4, 64, 64, 384, 384, 1), (4, 32, 32, 384, 384, 0), (4, 32, 32, 768, 384, 0), (1, 256, 256, 192, 192, 0), (1, 128, 128, 192, 192, 0), (1, 64, 64, 384, 384, 0), (2, 256, 256, 192, 192, 0), (2, 64, 64, 384, 384, 0), (8, 64, 64, 384, 384, 0)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for (N, H, W, C, Co, ups) in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    Hi, Wi = (H // 2, W // 2) if ups else (H, W)
    x = torch.randn((N, Hi, Wi, C), device=dev, generator=g); w = torch.randn((Co, C, 3, 3), device=dev, generator=g) * 0.02; b = torch.randn(Co, device=dev, generator=g)
    res = torch.randn((N, H, W, Co), device=dev, generator=g)
    cA = torch.rand((N, C), device=dev, generator=g) + 0.5; cB = torch.randn((N, C), device=dev, generator=g) * 0.1
    scratch = torch.empty(Co * C * 9 * 5 + 256 + (64 << 20) + N * Hi * Wi * C, device=dev)
    for gn in ((0, 1) if not ups else (0,)):
        outs, times = {}, {}
        for mode in ("0", "1"):
            os.environ["HL_WINO4W"] = mode
            out = torch.zeros((N, H, W, Co), device=dev)
            def call(r):
                _lib.check(L.hl_conv2d_nhwc_mode(0, _lib.ptr(x), N, Hi, Wi, C, _lib.ptr(w), _lib.ptr(b), Co, 3, 1, ups, _lib.ptr(cA) if gn else None,
                                                 _lib.ptr(cB) if gn else None, gn, _lib.ptr(r) if r is not None else None, _lib.ptr(out), _lib.ptr(scratch),
                                                 scratch.numel() * 4, _lib.stream_ptr()))
            call(res); torch.cuda.synchronize()
            outs[mode] = out.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): call(res)
            e1.record(); torch.cuda.synchronize()
            times[mode] = e0.elapsed_time(e1) / reps * 1e3
        d = float((outs["0"] - outs["1"]).abs().max())
        fl = 2.0 * N * H * W * Co * C * 9
        print(f"N{N} {H}x{W} C{C}->{Co} ups{ups} gn{gn}: max|wino4 - wino4w| = {d:.3e} (|out| max {float(outs['0'].abs().max()):.2f}, finite {bool(torch.isfinite(outs['1']).all())}); "
              f"wino4 {times['0']:.0f} us, wino4w {times['1']:.0f} us ({fl / times['1'] / 1e6:.0f} TF/s algorithmic; incl. weight packing + gn pass)", flush=True)
print("done")
