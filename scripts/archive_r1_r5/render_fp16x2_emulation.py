"""CPU emulation of the fp16x2 product arithmetic of k_march_plw<2> (two fp16 planes per operand, three partial products, fp32 accumulation) inside
the oracle renderer, against the REFERENCE golden renders (tests/golden/render_{a,b,c}.npz) - run before the kernel was written.
Modes: fp32 (the oracle as it is), rn_rn / rtz_rn / rtz_rtz = how the two activation planes are rounded (the kernel: rtz_rtz; weights: nearest even)."""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import render_oracle as ro
from tests.golden_util import load_render_case
import os
GOLDEN=None
def rn16(x): return x.to(torch.float16).to(torch.float32)
def rtz16(x):  # truncate mantissa to 10 bits (normal range)
    return (x.view(torch.int32) & ~0x1FFF).view(torch.float32)
def split(x, mode):
    if mode=='rn_rn': h0=rn16(x); h1=rn16(x-h0)
    elif mode=='rtz_rn': h0=rtz16(x); h1=rn16(x-h0)
    elif mode=='rtz_rtz': h0=rtz16(x); h1=rtz16(x-h0)
    return h0,h1
MODE='rtz_rn'; SX=1.0; SW=1.0
def lin(x, W, b):
    if MODE=='fp32': return F.linear(x, W, b)
    x0,x1=split(x*SX, MODE); w0=rn16(W*SW); w1=rn16(W*SW-w0)
    # three products, fp32 accumulation (order: small first)
    acc = (x1@w0.t()) + (x0@w1.t())
    acc = acc + (x0@w0.t())
    return acc/(SX*SW) + b
def mlp(p, feats, dirs=None):
    h = F.softplus(lin(feats, p["pts_linears.0.weight"], p["pts_linears.0.bias"]))
    h = F.softplus(lin(h, p["pts_linears.1.weight"], p["pts_linears.1.bias"]))
    h = torch.cat([feats, h], -1)
    h = F.softplus(lin(h, p["pts_linears.2.weight"], p["pts_linears.2.bias"]))
    sigma = F.linear(h, p["alpha_linear.weight"], p["alpha_linear.bias"])[:, 0]
    if dirs is None: return sigma
    feat = lin(h, p["feature_linear.weight"], p["feature_linear.bias"])
    h = torch.cat([feat, ro.view_encoding(dirs)], -1)
    h = F.softplus(lin(h, p["views_linear.weight"], p["views_linear.bias"]))
    rgb = F.linear(h, p["rgb_linear.weight"], p["rgb_linear.bias"])
    return rgb, sigma
import inspect
print(inspect.signature(ro.render_rays))
orig = ro.mlp
for name in ("render_a","render_b","render_c"):
    inp, e = load_render_case(name[-1]); mlp_p = inp["mlp"]
    for MODE in ('fp32','rn_rn','rtz_rn','rtz_rtz'):
        for (SX,SW) in ((1.0,1.0),(4.0,16.0)):
            if MODE=='fp32' and SX!=1.0: continue
            ro.mlp = mlp
            out = ro.render_rays(mlp_p, inp["planes"][0], inp["bounds"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], int(inp["n_samples"]), int(inp["n_importance"]), u=inp["u"], white_bkgd=bool(inp.get("white_bkgd", False)))
            rgb = out[0] if isinstance(out,(tuple,list)) else out["rgb"]
            print(name, MODE, SX, SW, 'rgb max-abs vs reference golden: %.3e'%float((rgb - e["rgb"]).abs().max()))
ro.mlp = orig
