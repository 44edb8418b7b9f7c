"""Developer probe: k_conv_h16 (16-bit operands, fp32 accumulation) against the fp32 kernels and against a float64 convolution of the
ROUNDED operands (what the kernel should compute exactly up to fp32 summation order); event-timed (incl. weight packing)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
shapes = [(1, 32, 32, 32, 192, 0), (2, 64, 64, 96, 192, 1), (2, 48, 80, 64, 384, 1), (4, 256, 256, 192, 192, 1), (4, 256, 256, 384, 192, 0), (4, 128, 128, 192, 192, 0), (4, 64, 64, 384, 384, 0)]
for (N, H, W, C, Co, use_res) in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, W, C), device=dev, generator=g); w = torch.randn((Co, C, 3, 3), device=dev, generator=g) / (C * 9) ** 0.5; b = torch.randn(Co, device=dev, generator=g)
    res = torch.randn((N, H, W, Co), device=dev, generator=g) if use_res else None
    scratch = torch.empty(Co * C * 9 * 6 + 256 + (64 << 20), device=dev)
    outs = {}
    for mode in (0, 4, 5):
        out = torch.zeros((N, H, W, Co), device=dev)
        def call():
            _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, 3, 1, 0, None, None, 0, _lib.ptr(res) if use_res else None,
                                             _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        outs[mode] = (out.clone(), e0.elapsed_time(e1) / 10 * 1e3)
    msg = []
    if N * H * W <= 2 * 64 * 80:
        for mode, dt in ((4, torch.bfloat16), (5, torch.float16)):
            xr, wr = x.to(dt).double(), w.to(dt).double()
            ref = F.conv2d(xr.permute(0, 3, 1, 2), wr, b.double(), padding=1).permute(0, 2, 3, 1)
            if use_res: ref = ref + res.double()
            msg.append(f"mode {mode} vs float64 conv of rounded operands: max-abs {float((outs[mode][0].double() - ref).abs().max()):.2e}")
    ref = outs[0][0].double()
    fl = 2.0 * N * H * W * Co * C * 9
    print(f"N{N} {H}x{W} {C}->{Co} res{use_res}: " + "; ".join(f"mode {m}: {t:.0f} us ({fl / t / 1e6:.0f} TF/s), rel-L2 vs fp32 {float((o.double() - ref).norm() / ref.norm()):.2e}" for m, (o, t) in outs.items()) + " | " + "; ".join(msg), flush=True)

for (N, H, W, C, Co) in [(4, 256, 256, 384, 192), (4, 64, 64, 384, 1152), (4, 128, 128, 576, 192), (4, 64, 64, 384, 384)]:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, W, C), device=dev, generator=g); w = torch.randn((Co, C, 1, 1), device=dev, generator=g) / C ** 0.5; b = torch.randn(Co, device=dev, generator=g)
    scratch = torch.empty(Co * C * 6 + 256 + (64 << 20), device=dev)
    msg = []
    for mode in (0, 4, 5):
        out = torch.zeros((N, H, W, Co), device=dev)
        def call():
            _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, 1, 1, 0, None, None, 0, None, _lib.ptr(out), _lib.ptr(scratch),
                                             scratch.numel() * 4, _lib.stream_ptr()))
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e3
        msg.append(f"mode {mode}: {t:.0f} us ({2.0 * N * H * W * C * Co / t / 1e6:.0f} TF/s, {(N * H * W * (C + Co) * 4) / t / 1e6:.2f} TB/s)")
    print(f"1x1 N{N} {H}x{W} {C}->{Co}: " + "; ".join(msg), flush=True)
