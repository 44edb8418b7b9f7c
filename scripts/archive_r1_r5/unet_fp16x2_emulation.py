"""CPU emulation: the 1x1 convolutions of the production UNet (ResBlock skips, attention qkv / proj_out, the control tower's zero convolutions) with
fp16x2 products - two fp16 planes per operand (activations truncated / truncated, weights nearest-even), three partial products, fp32 accumulation -
inside the oracle, against the plain fp32 oracle.  Decides whether k_conv1_h16 may take these layers in the DEFAULT (fp32-tolerance) mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
from humanliff_amd import synthetic as syn
from oracle import unet_oracle as uo

def rtz16(x): return (x.view(torch.int32) & ~0x1FFF).view(torch.float32)
def rn16(x): return x.to(torch.float16).to(torch.float32)
def split_act(x):
    h0 = rtz16(x); return h0, rtz16(x - h0)
def split_w(w):
    w0 = rn16(w); return w0, rn16(w - w0)
orig2d, orig1d = F.conv2d, F.conv1d
def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
    if w.shape[-1] == 1 and w.shape[-2] == 1 and stride == 1:
        x0, x1 = split_act(x); w0, w1 = split_w(w)
        y = orig2d(x1, w0) + orig2d(x0, w1)
        y = y + orig2d(x0, w0)
        return y if b is None else y + b.view(1, -1, 1, 1)
    return orig2d(x, w, b, stride, padding, *a, **k)
def conv1d(x, w, b=None, *a, **k):
    x0, x1 = split_act(x); w0, w1 = split_w(w)
    y = orig1d(x1, w0) + orig1d(x0, w1)
    y = y + orig1d(x0, w0)
    return y if b is None else y + b.view(1, -1, 1)
torch.set_num_threads(8)
from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion
model, _ = create_model_and_diffusion(**bench.F4)
sd = syn.state_from_shapes([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=1)
del model
g = torch.Generator().manual_seed(123)
x = torch.randn((1, 27, 256, 256), generator=g); xc = torch.randn((1, 27, 256, 256), generator=g).clamp(-1, 1) * 0.7
t = torch.tensor([617]); y = torch.tensor([2])
with torch.no_grad():
    t0 = time.time(); want = uo.unet_forward(sd, x, t, xc, y, num_heads=4); print("fp32 oracle", time.time() - t0, "s")
    F.conv2d, F.conv1d = conv2d, conv1d
    try:
        got = uo.unet_forward(sd, x, t, xc, y, num_heads=4)
    finally:
        F.conv2d, F.conv1d = orig2d, orig1d
scale = float(want.abs().mean())
print(f"1x1 convolutions as fp16x2 products: max-abs vs the fp32 oracle {float((got - want).abs().max()):.3e} (output mean-abs {scale:.3f}; bound of the parity tests 5e-5 x scale = {5e-5 * max(1, scale):.1e})")
