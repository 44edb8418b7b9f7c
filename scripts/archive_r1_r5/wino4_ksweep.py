"""Developer probe: k_conv_wino4 launch time against the K length at 256x256, batch 4, 192 output channels (rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
N, H, W, Co = 4, 256, 256, 192
for C in (16, 48, 96, 192, 384, 768):
    x = torch.randn((N, H, W, C), device=dev); w = torch.randn((Co, C, 3, 3), device=dev) * 0.02; b = torch.randn(Co, device=dev)
    out = torch.empty((N, H, W, Co), device=dev); scratch = torch.empty(Co * C * 9 * 5 + 256 + (16 << 20), device=dev)
    for rep in range(4):
        _lib.check(L.hl_conv2d_nhwc_mode(0, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, 3, 1, 0, None, None, 0, None,
                                         _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
print("done")
