"""Developer probe: k_conv_wino launch time against the K length (input channels) at the dominant shape (256x256, batch 4, 192
output channels).  Run under rocprofv3 --kernel-trace and list the dispatches with scripts/rocpd_list.py:
    rocprofv3 --kernel-trace -d out -- python scripts/wino_k_sweep.py ; python scripts/rocpd_list.py out k_conv_wino"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
N, H, W, Co = 4, 256, 256, 192
for C in (48, 96, 192, 384, 768):
    x = torch.randn((N, H, W, C), device=dev); w = torch.randn((Co, C, 3, 3), device=dev) * 0.02; b = torch.randn(Co, device=dev)
    res = torch.randn((N, H, W, Co), device=dev)
    out = torch.empty((N, H, W, Co), device=dev); scratch = torch.empty(256 * C * 9 * 5 + 256 + (4 << 20), device=dev)
    for rep in range(3):
        for r in (None, res):
            _lib.check(L.hl_conv2d_nhwc_mode(3, _lib.ptr(x), N, H, W, C, _lib.ptr(w), _lib.ptr(b), Co, 3, 1, 0, None, None, 0, _lib.ptr(r) if r is not None else None,
                                             _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
print("done")
