"""Developer probe: the 1x1 layers of the default mode through the single-op entry point (HL_B, default 4), 5 calls each - run under rocprofv3 --kernel-trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
B = int(os.environ.get("HL_B", "4"))
shapes = [(B, 256, 256, 384, 192), (B, 128, 128, 576, 192), (B, 64, 64, 768, 384), (B, 32, 32, 576, 1728)]
for (N, H, W, C, Co) in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, W, C), device=dev, generator=g); w = torch.randn((Co, C, 1, 1), device=dev, generator=g) / C ** 0.5; b = torch.randn(Co, device=dev, generator=g)
    res = torch.randn((N, H, W, Co), device=dev, generator=g) if os.environ.get("HL_RES") else None
    scratch = torch.empty(Co * C * 8 + 256 + (64 << 20), device=dev)
    out = torch.zeros((N, H, W, Co), device=dev)
    for it in range(6):
        _lib.check(L.hl_conv2d_nhwc_mode(0, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, 1, 1, 0, None, None, 0, _lib.ptr(res) if res is not None else None, _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
    print(f"N{N} {H}x{W} {C}->{Co}", flush=True)
