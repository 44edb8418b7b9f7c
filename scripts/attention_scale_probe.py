"""Developer probe: relative error of the attention (fp16x2 products, the default mode; fp32 MFMA for comparison) against float64 when V, or Q and K, are scaled by powers of two."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from humanliff_amd import _lib
dev = torch.device("cuda:0")
N, T, C, heads = 2, 1024, 384, 4
ch = C // heads
g = torch.Generator().manual_seed(3)
base = torch.randn((N, 3 * C, T), generator=g, dtype=torch.float64)
for name, sq, sv in (("as drawn", 1.0, 1.0), ("V x 2^-6", 1.0, 2.0 ** -6), ("V x 2^-10", 1.0, 2.0 ** -10), ("V x 2^-14", 1.0, 2.0 ** -14), ("V x 2^10", 1.0, 2.0 ** 10),
                     ("Q, K x 2^-3", 0.125, 1.0), ("Q, K x 2^-6", 2.0 ** -6, 1.0), ("all x 2^-6", 2.0 ** -6, 2.0 ** -6)):
    qkv = base.clone().reshape(N * heads, 3 * ch, T)
    qkv[:, :2 * ch] *= sq
    qkv[:, 2 * ch:] *= sv
    q, k, v = qkv.split(ch, dim=1)
    s = 1.0 / (ch ** 0.25)
    want = torch.einsum("bts,bcs->bct", torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1), v).reshape(N, C, T)
    qd = qkv.reshape(N, 3 * C, T).permute(0, 2, 1).contiguous().float().to(dev)
    res = []
    for h2 in (True, False):
        out = torch.empty((N, T, C), device=dev)
        if h2:
            _lib.check(_lib.lib().hl_attention_nhwc_mode(_lib.HL_CONV_FP32, _lib.ptr(qd), N, T, C, heads, _lib.ptr(out), _lib.stream_ptr()))
        else:
            _lib.check(_lib.lib().hl_attention_nhwc(_lib.ptr(qd), N, T, C, heads, _lib.ptr(out), _lib.stream_ptr()))
        torch.cuda.synchronize()
        d = out.cpu().double().permute(0, 2, 1) - want
        res.append(float(d.norm() / want.norm()))
    print(f"{name:14s} rel-L2 vs float64: fp16x2 {res[0]:.2e}   fp32 MFMA {res[1]:.2e}", flush=True)
