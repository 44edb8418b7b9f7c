"""Developer probe: k_conv_wino4 (F(4x4,3x3)) against k_conv_wino (F(2x2,3x3)) on the layer shapes of the production UNet at batch 4.
Run under rocprofv3 --kernel-trace and list the dispatches with scripts/rocpd_list.py:
    rocprofv3 --kernel-trace -d out -- python scripts/wino4_probe.py ; python scripts/rocpd_list.py out k_conv_wino"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
shapes = [(4, 256, 256, 192, 192), (4, 256, 256, 384, 192), (4, 128, 128, 192, 192), (4, 128, 128, 384, 192), (4, 64, 64, 384, 384), (4, 64, 64, 768, 384)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (N, H, W, C, Co) in shapes:
    x = torch.randn((N, H, W, C), device=dev); w = torch.randn((Co, C, 3, 3), device=dev) * 0.02; b = torch.randn(Co, device=dev)
    res = torch.randn((N, H, W, Co), device=dev)
    out = torch.empty((N, H, W, Co), device=dev); scratch = torch.empty(Co * C * 9 * 5 + 256 + (64 << 20), device=dev)
    for mode in (3, 0):   # HL_CONV_FP32_F23 (F(2x2) only), HL_CONV_FP32 (F(4x4) where it fills the chip)
        for rep in range(3):
            for r in (None, res):
                _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, 3, 1, 0, None, None, 0,
                                                 _lib.ptr(r) if r is not None else None, _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4,
                                                 _lib.stream_ptr()))
        torch.cuda.synchronize()
print("done")
