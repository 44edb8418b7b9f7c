"""Developer probe: the split-K Winograd F(2x2,3x3) launches of the 32- and 16-pixel levels (batch 4) under rocprofv3 --kernel-trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
for (N, H, W, C, Co) in [(4, 32, 32, 384, 384), (4, 32, 32, 768, 384), (4, 16, 16, 768, 768), (4, 16, 16, 1536, 768), (4, 64, 64, 384, 384)]:
    x = torch.randn((N, H, W, C), device=dev); w = torch.randn((Co, C, 3, 3), device=dev) * 0.02; b = torch.randn(Co, device=dev)
    out = torch.empty((N, H, W, Co), device=dev); scratch = torch.empty(Co * C * 9 * 5 + 256 + (64 << 20), device=dev)
    for rep in range(4):
        _lib.check(L.hl_conv2d_nhwc_mode(0, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, 3, 1, 0, None, None, 0, None,
                                         _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
print("done")
