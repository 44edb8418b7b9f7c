"""Developer probe: k_conv_wino4 behind a GroupNorm+SiLU prologue (the channel-blocked copy written by k_gn_apply_blk) at the 256x256 and
128x128 shapes of the production UNet, batch 4 (rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
for (N, H, W, C, Co) in [(4, 256, 256, 192, 192), (4, 256, 256, 384, 192), (4, 128, 128, 192, 192)]:
    x = torch.randn((N, H, W, C), device=dev); w = torch.randn((Co, C, 3, 3), device=dev) * 0.02; b = torch.randn(Co, device=dev)
    cA = torch.rand((N, C), device=dev) + 0.5; cB = torch.randn((N, C), device=dev) * 0.3
    out = torch.empty((N, H, W, Co), device=dev); scratch = torch.empty(Co * C * 9 * 5 + 256 + (64 << 20) + N * H * W * C, device=dev)
    for rep in range(4):
        _lib.check(L.hl_conv2d_nhwc_mode(0, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, 3, 1, 0, _lib.ptr(cA), _lib.ptr(cB), 1, None,
                                         _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
print("done")
