"""GPU parity of the UNet path (through the C ABI): single ops vs torch CPU fp32, whole model vs the
reference's golden outputs, sampler steps bit-exact vs the oracle, full loops vs the reference."""
import os

import numpy as np
import pytest
import torch
from tests.unet_autograd_twin import forward_autograd
import torch.nn.functional as F

from tests.golden_util import GOLDEN
from tests.test_oracle_diffusion import load_unet_case, _stub

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def hip_conv(x, w, b, ks, stride=1, ups=0, cA=None, cB=None, silu=0, res=None, mode=0):
    from humanliff_amd import _lib
    L = _lib.lib()
    N, C, H, W = x.shape
    Cout = w.shape[0]
    xin = nhwc(x).to(dev)
    Hv, Wv = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = (Hv + 2 * (ks // 2) - ks) // stride + 1, (Wv + 2 * (ks // 2) - ks) // stride + 1
    out = torch.empty((N, Ho, Wo, Cout), device=dev)
    # packed weights + room for split-K partial sums (taken for shapes that would under-fill the chip)
    scratch = torch.empty(((Cout + 63) // 64 * 64) * C * ks * ks * 5 + 256 + (8 << 20) + N * C * H * W, device=dev)
    d = lambda t: None if t is None else t.contiguous().to(dev)  # noqa: E731
    wd, bd, cAd, cBd = d(w), d(b), d(cA), d(cB)
    rd = d(nhwc(res)) if res is not None else None
    _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(xin), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Cout, ks, stride, ups,
                                     _lib.ptr(cAd), _lib.ptr(cBd), silu, _lib.ptr(rd), _lib.ptr(out), _lib.ptr(scratch),
                                     scratch.numel() * 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return nchw(out.cpu())


@pytest.mark.parametrize("N,C,H,W,Cout,ks,stride,ups", [
    (2, 32, 16, 16, 64, 3, 1, 0),      # small-tile config
    (1, 192, 32, 32, 192, 3, 1, 0),    # production channel count
    (4, 192, 64, 64, 192, 3, 1, 0),    # M=16384
    (2, 192, 128, 128, 192, 3, 1, 0),  # main tile config (128x96, 4 waves/SIMD), 512 workgroups, XCD remap
    (1, 96, 160, 96, 96, 3, 1, 0),     # main tile config with a ragged last pixel tile (M=15360) and Cout=96
    (2, 64, 16, 16, 64, 3, 2, 0),      # Downsample (unet.py:100)
    (2, 64, 8, 8, 64, 3, 1, 1),        # Upsample: nearest x2 then conv (unet.py:77-79)
    (2, 96, 12, 20, 128, 1, 1, 0),     # 1x1 skip / zero-conv, ragged M
    (1, 64, 16, 16, 27, 3, 1, 0),      # 27 output channels (N tile 32)
    (2, 384, 8, 8, 1152, 1, 1, 0),     # qkv projection (split-K x3)
    (4, 768, 8, 8, 768, 3, 1, 0),      # 8x8 level of the production UNet: K=6912, split-K
])
def test_conv_matches_torch(N, C, H, W, Cout, ks, stride, ups):
    g = torch.Generator().manual_seed(N * 1000 + C + H + Cout)
    x = torch.randn((N, C, H, W), generator=g)
    w = torch.randn((Cout, C, ks, ks), generator=g) / (C * ks * ks) ** 0.5
    b = torch.randn((Cout,), generator=g)
    got = hip_conv(x, w, b, ks, stride, ups)
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    want = F.conv2d(xi, w, b, stride=stride, padding=ks // 2)
    assert got.shape == want.shape
    assert (got - want).abs().max() < 2e-5          # outputs O(1), K up to 3456: fp32 accumulation-order noise


@pytest.mark.parametrize("N,C,H,W,Cout,ks,stride,ups", [
    (1, 192, 32, 32, 192, 3, 1, 0),    # 128x96 tile, split-K
    (2, 192, 128, 128, 192, 3, 1, 0),  # 256x96 tile (8 waves)
    (1, 96, 160, 96, 96, 3, 1, 0),     # ragged last pixel tile, Cout=96
    (2, 96, 16, 16, 96, 3, 2, 0),      # stride 2
    (2, 96, 24, 24, 96, 3, 1, 1),      # nearest x2 upsample
    (2, 384, 8, 8, 1152, 1, 1, 0),     # 1x1 (qkv), split-K
    (4, 768, 8, 8, 768, 3, 1, 0),      # K = 6912, 16 slabs
])
def test_conv_bf16x3_emulation_matches_torch(N, C, H, W, Cout, ks, stride, ups):
    """HL_CONV_BF16X3: fp32 products emulated with three bf16 planes per operand - same bound as the exact fp32 kernel,
    and the difference between the two modes is at accumulation-noise level."""
    from humanliff_amd import _lib
    g = torch.Generator().manual_seed(N * 1000 + C + H + Cout)
    x = torch.randn((N, C, H, W), generator=g)
    w = torch.randn((Cout, C, ks, ks), generator=g) / (C * ks * ks) ** 0.5
    b = torch.randn((Cout,), generator=g)
    got = hip_conv(x, w, b, ks, stride, ups, mode=_lib.HL_CONV_BF16X3)
    exact = hip_conv(x, w, b, ks, stride, ups, mode=_lib.HL_CONV_FP32)
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    want = F.conv2d(xi.double(), w.double(), b.double(), stride=stride, padding=ks // 2)
    e_bf3, e_f32 = (got.double() - want).abs().max().item(), (exact.double() - want).abs().max().item()
    assert e_bf3 < 2e-5                      # the fp32 kernel's bound (test_conv_matches_torch)
    assert e_bf3 < 4 * e_f32 + 1e-6          # and within a small factor of the exact kernel's error vs float64


def test_conv_winograd_upsample_matches_torch():
    """nearest x2 + 3x3 conv (Upsample, unet.py:77-79) through the Winograd kernel: the patch DMA reads source pixel (y>>1, x>>1)."""
    from humanliff_amd import _lib
    g = torch.Generator().manual_seed(77)
    N, C, H, W, Cout = 4, 96, 64, 48, 192
    x = torch.randn((N, C, H, W), generator=g)
    w = torch.randn((Cout, C, 3, 3), generator=g) / (C * 9) ** 0.5
    b = torch.randn((Cout,), generator=g)
    got = hip_conv(x, w, b, 3, ups=1, mode=_lib.HL_CONV_FP32_F23)
    direct = hip_conv(x, w, b, 3, ups=1, mode=_lib.HL_CONV_FP32_DIRECT)
    want = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    assert got.shape == want.shape and not torch.equal(got, direct)
    e_w, e_d = (got.double() - want).abs().max().item(), (direct.double() - want).abs().max().item()
    assert e_w < 2e-5 and e_w < 8 * e_d + 1e-6, (e_w, e_d)



@pytest.mark.parametrize("N,C,H,W,Cout,with_gn", [
    (4, 32, 128, 128, 192, False),    # 768 workgroups, 4 k-tiles
    (1, 192, 256, 256, 192, False),   # production shape (batch 1): 24 k-tiles
    (3, 96, 176, 208, 192, False),    # non-square, H, W multiples of 16 only
    (2, 64, 256, 128, 192, True),     # GroupNorm+SiLU prologue (materialised), residual epilogue
    (9, 48, 64, 64, 384, False),      # 6 channel blocks
    (4, 384, 32, 32, 384, True),      # 192 workgroups: input channels split into 3 slabs, summed by k_splitk_finish
    (4, 768, 16, 16, 768, False),     # 96 workgroups x 6 slabs
])
def test_conv_winograd_matches_torch(N, C, H, W, Cout, with_gn):
    """HL_CONV_FP32 takes Winograd F(2x2,3x3) for large 3x3 layers: fp32 arithmetic, 2.25x fewer multiplies; compared with
    float64 torch and with the direct kernel (HL_CONV_FP32_DIRECT) on the same inputs."""
    from humanliff_amd import _lib
    g = torch.Generator().manual_seed(N * 1000 + C + H + Cout)
    x = torch.randn((N, C, H, W), generator=g)
    w = torch.randn((Cout, C, 3, 3), generator=g) / (C * 9) ** 0.5
    b = torch.randn((Cout,), generator=g)
    kw = {}
    xin = x.double()
    if with_gn:
        cA, cB = torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g) * 0.3
        res = torch.randn((N, Cout, H, W), generator=g)
        kw = dict(cA=cA, cB=cB, silu=1, res=res)
        hn = x.double() * cA.double()[:, :, None, None] + cB.double()[:, :, None, None]
        xin = hn * torch.sigmoid(hn)
    got = hip_conv(x, w, b, 3, mode=_lib.HL_CONV_FP32_F23, **kw)
    direct = hip_conv(x, w, b, 3, mode=_lib.HL_CONV_FP32_DIRECT, **kw)
    want = F.conv2d(xin, w.double(), b.double(), padding=1)
    if with_gn:
        want = want + res.double()
    assert not torch.equal(got, direct)                      # the Winograd kernel really ran
    e_w, e_d = (got.double() - want).abs().max().item(), (direct.double() - want).abs().max().item()
    assert e_w < 2e-5, (e_w, e_d)                            # the direct kernel's bound (outputs O(1))
    assert e_w < 8 * e_d + 1e-6, (e_w, e_d)                  # F(2x2,3x3) amplifies rounding by a small constant


@pytest.mark.parametrize("N,C,H,W,Cout,with_gn,ups", [
    (4, 32, 128, 128, 192, False, 0),   # 768 workgroups, 4 k-tiles
    (1, 192, 256, 256, 192, False, 0),  # production shape (batch 1): 24 k-tiles
    (3, 96, 176, 224, 192, False, 0),   # non-square: H a multiple of 16, W of 32
    (2, 64, 256, 128, 192, True, 0),    # GroupNorm+SiLU prologue (materialised), residual epilogue
    (9, 48, 64, 64, 384, False, 0),     # 12 channel blocks, odd k-tile count (6) and batch
    (2, 16, 128, 256, 192, False, 0),   # two k-tiles
    (4, 96, 64, 48, 192, False, 1),     # nearest x2 upsample in front (the patch DMA reads source pixel (y>>1, x>>1))
    # the three dispatch regimes of k_conv_wino4w (64-channel workgroups, one per CU; W = N x H/16 x W/32 x Cout/64 workgroups):
    (4, 32, 256, 256, 192, False, 0),   # W = 1536 >= 768: three rounds and more (the 256-pixel level at batch >= 2)
    (4, 64, 64, 64, 384, True, 0),      # W = 192 in [160, 256]: one nearly full round (the 64-pixel level at batch 4), GroupNorm prologue + residual
    (1, 384, 64, 64, 384, False, 0),    # W = 48 < 160: input channels split into slabs, summed by k_splitk_finish (the 64-pixel level at batch 1)
])
def test_conv_winograd_f43_matches_torch(N, C, H, W, Cout, with_gn, ups):
    """HL_CONV_FP32 takes Winograd F(4x4,3x3) with the points (0, +-3/4, +-3/2, inf) where it fills the chip - a quarter of the direct
    multiplies, fp32 throughout; compared with float64 torch, with the F(2x2) kernel and with the direct kernel on the same inputs."""
    from humanliff_amd import _lib
    g = torch.Generator().manual_seed(N * 1000 + C + H + Cout)
    x = torch.randn((N, C, H, W), generator=g)
    w = torch.randn((Cout, C, 3, 3), generator=g) / (C * 9) ** 0.5
    b = torch.randn((Cout,), generator=g)
    kw = {}
    xin = x.double()
    if with_gn:
        cA, cB = torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g) * 0.3
        res = torch.randn((N, Cout, H, W), generator=g)
        kw = dict(cA=cA, cB=cB, silu=1, res=res)
        hn = x.double() * cA.double()[:, :, None, None] + cB.double()[:, :, None, None]
        xin = hn * torch.sigmoid(hn)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    got = hip_conv(x, w, b, 3, ups=ups, mode=_lib.HL_CONV_FP32, **kw)
    f23 = hip_conv(x, w, b, 3, ups=ups, mode=_lib.HL_CONV_FP32_F23, **kw)
    direct = hip_conv(x, w, b, 3, ups=ups, mode=_lib.HL_CONV_FP32_DIRECT, **kw)
    want = F.conv2d(xin, w.double(), b.double(), padding=1)
    if with_gn:
        want = want + res.double()
    assert not torch.equal(got, direct) and not torch.equal(got, f23)   # the F(4x4) kernel really ran
    e_w, e_d = (got.double() - want).abs().max().item(), (direct.double() - want).abs().max().item()
    r_w, r_d = (got.double() - want).pow(2).mean().sqrt().item(), (direct.double() - want).pow(2).mean().sqrt().item()
    assert e_w < 4e-5, (e_w, e_d)                            # outputs O(1)
    assert r_w < 8 * r_d + 1e-7, (r_w, r_d)                  # rms error within a small constant of the direct kernel's


def test_conv_fused_groupnorm_silu_residual():
    """GroupNorm(+scale/shift) -> SiLU -> conv3x3 + bias + residual, the ResBlock inner step (unet.py:198-219)."""
    from humanliff_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    N, C, H, W, Cout = 2, 96, 24, 24, 96
    x = torch.randn((N, C, H, W), generator=g) * 2 + 0.5
    gamma, beta = torch.randn(C, generator=g) * 0.2 + 1, torch.randn(C, generator=g) * 0.2
    emb = torch.randn((N, 2 * C), generator=g) * 0.3
    w = torch.randn((Cout, C, 3, 3), generator=g) / (C * 9) ** 0.5
    b = torch.randn((Cout,), generator=g)
    res = torch.randn((N, Cout, H, W), generator=g)
    xd = nhwc(x).to(dev)
    cA, cB = torch.empty((N, C), device=dev), torch.empty((N, C), device=dev)
    scratch = torch.empty(N * 128 * 64 + 64, device=dev)
    gd_, bd_, ed_ = gamma.to(dev), beta.to(dev), emb.to(dev)
    _lib.check(L.hl_groupnorm_coef(_lib.ptr(xd), N, H, W, C, _lib.ptr(gd_), _lib.ptr(bd_), _lib.ptr(ed_), _lib.ptr(cA),
                                   _lib.ptr(cB), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
    hn = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    scale, shift = emb.chunk(2, dim=1)
    hn2 = hn * (1 + scale[:, :, None, None]) + shift[:, :, None, None]
    # the affine reproduces GN*(1+scale)+shift
    rec = x * cA.cpu()[:, :, None, None] + cB.cpu()[:, :, None, None]
    assert (rec - hn2).abs().max() < 2e-5
    got = hip_conv(x, w, b, 3, cA=cA.cpu(), cB=cB.cpu(), silu=1, res=res)
    want = F.conv2d(hn2 * torch.sigmoid(hn2), w, b, padding=1) + res
    assert (got - want).abs().max() < 3e-5


@pytest.mark.parametrize("N,C,H,W,use_emb", [(2, 384, 16, 16, True), (3, 192, 8, 8, False), (1, 96, 12, 20, True),
                                              (2, 1152, 8, 8, True), (2, 192, 64, 64, True), (1, 64, 128, 96, False)])
def test_groupnorm_affine_matches_torch(N, C, H, W, use_emb):
    """GroupNorm32 statistics folded to a per-(n,c) affine (nn.py:17-19,100; scale/shift of unet.py:212-216):
    small tensors take the one-workgroup-per-group kernel, large ones the chunked two-pass."""
    from humanliff_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn((N, C, H, W), generator=g) * 1.5 + 0.3
    gamma, beta = torch.randn(C, generator=g) * 0.2 + 1, torch.randn(C, generator=g) * 0.2
    emb = torch.randn((N, 2 * C), generator=g) * 0.3
    xd, gd_, bd_, ed_ = nhwc(x).to(dev), gamma.to(dev), beta.to(dev), emb.to(dev)
    cA, cB = torch.empty((N, C), device=dev), torch.empty((N, C), device=dev)
    scratch = torch.empty(N * 128 * 64 + 64, device=dev)
    _lib.check(L.hl_groupnorm_coef(_lib.ptr(xd), N, H, W, C, _lib.ptr(gd_), _lib.ptr(bd_), _lib.ptr(ed_) if use_emb else None,
                                   _lib.ptr(cA), _lib.ptr(cB), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
    want = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    if use_emb:
        scale, shift = emb.chunk(2, dim=1)
        want = want * (1 + scale[:, :, None, None]) + shift[:, :, None, None]
    rec = x * cA.cpu()[:, :, None, None] + cB.cpu()[:, :, None, None]
    assert (rec - want).abs().max() < 2e-5


@pytest.mark.parametrize("N,T,C,heads", [(2, 64, 128, 4), (1, 256, 384, 4), (2, 1024, 384, 4), (1, 64, 768, 4), (1, 100, 64, 4),
                                          (1, 100, 384, 4), (2, 200, 768, 4), (4, 1024, 384, 4), (16, 1024, 384, 4)])
@pytest.mark.parametrize("h2", [False, True])
def test_attention_matches_torch(N, T, C, heads, h2):
    """h2: the attention of the default conv mode (hl_attention_nhwc_mode(HL_CONV_FP32)): scores and probabilities x values from fp16x2 operands where the kernel has that
    form (head sizes 96 / 192 on short sequences), same bound."""
    from humanliff_amd import _lib
    g = torch.Generator().manual_seed(T + C)
    qkv = torch.randn((N, 3 * C, T), generator=g)          # reference layout (b, 3C, T)
    ch = C // heads
    q, k, v = qkv.reshape(N * heads, 3 * ch, T).split(ch, dim=1)
    s = 1.0 / (ch ** 0.25)
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
    want = torch.einsum("bts,bcs->bct", wgt, v).reshape(N, C, T)
    qd = qkv.permute(0, 2, 1).contiguous().to(dev)          # (N, T, 3C)
    out = torch.empty((N, T, C), device=dev)
    if h2:
        _lib.check(_lib.lib().hl_attention_nhwc_mode(_lib.HL_CONV_FP32, _lib.ptr(qd), N, T, C, heads, _lib.ptr(out), _lib.stream_ptr()))
    else:
        _lib.check(_lib.lib().hl_attention_nhwc(_lib.ptr(qd), N, T, C, heads, _lib.ptr(out), _lib.stream_ptr()))
    torch.cuda.synchronize()
    err = float((out.cpu().permute(0, 2, 1) - want).abs().max())
    print(f"attention N{N} T{T} C{C} h2={h2}: max-abs {err:.2e}")
    assert err < 2e-5


@pytest.mark.parametrize("N,T,C,heads", [(2, 1024, 384, 4), (1, 256, 768, 4)])
def test_attention_fp16x2_is_invariant_to_the_scale_of_v(N, T, C, heads):
    """Round 6 ("the attention's K/V side"): the fp16x2 attention stores its V tiles multiplied by a running power of two (k_attention_ks<., ., ., true>),
    so the two fp16 planes of V stay normal whatever V's magnitude.  Scaling V by 2^k scales the exact result by 2^k: the kernel's output must follow BIT FOR BIT from
    2^-14 to 2^10 (without the scale: rel-L2 against float64 3.6e-7 as drawn, 1.8e-5 at 2^-10, 2.8e-4 at 2^-14), and stay at or below the fp32-MFMA kernel's error."""
    from humanliff_amd import _lib
    g = torch.Generator().manual_seed(T + C + 1)
    base = torch.randn((N, 3 * C, T), generator=g)
    ch = C // heads
    s = 1.0 / (ch ** 0.25)

    def run(qkv, h2):
        qd = qkv.permute(0, 2, 1).contiguous().to(dev)
        out = torch.empty((N, T, C), device=dev)
        if h2:
            _lib.check(_lib.lib().hl_attention_nhwc_mode(_lib.HL_CONV_FP32, _lib.ptr(qd), N, T, C, heads, _lib.ptr(out), _lib.stream_ptr()))
        else:
            _lib.check(_lib.lib().hl_attention_nhwc(_lib.ptr(qd), N, T, C, heads, _lib.ptr(out), _lib.stream_ptr()))
        torch.cuda.synchronize()
        return out.cpu().permute(0, 2, 1)

    q, k, v = base.double().reshape(N * heads, 3 * ch, T).split(ch, dim=1)
    want = torch.einsum("bts,bcs->bct", torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1), v).reshape(N, C, T)
    ref = run(base, True)
    e2 = float((ref.double() - want).norm() / want.norm())
    e32 = float((run(base, False).double() - want).norm() / want.norm())
    print(f"attention T{T} C{C}: rel-L2 against float64 fp16x2 {e2:.2e}, fp32 MFMA {e32:.2e}")
    assert e2 < 1.3 * e32 + 5e-8
    for kexp in (-14, -10, -6, 5, 10):
        scaled = base.clone().reshape(N * heads, 3 * ch, T)
        scaled[:, 2 * ch:] *= 2.0 ** kexp
        got = run(scaled.reshape(N, 3 * C, T), True)
        assert torch.equal(got * 2.0 ** -kexp, ref), kexp


def test_timestep_embedding_matches_reference():
    from humanliff_amd.improved_diffusion.nn import timestep_embedding
    g = np.load(os.path.join(GOLDEN, "diffusion_steps.npz"))
    got = timestep_embedding(torch.tensor([0, 1, 500, 999], device=dev), 192).cpu()
    assert (got - torch.from_numpy(g["temb192"])).abs().max() < 2e-5   # |arg| up to 999 rad: sin/cos of fp32 args


def build_model(g, sd):
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults
    a = model_and_diffusion_defaults()
    a.update(dict(in_channels=27, out_channels=27, class_cond=True, num_heads=int(g["arg_num_heads"]), rescale_timesteps=False,
                  image_size=int(g["arg_image_size"]), num_channels=int(g["arg_num_channels"]),
                  num_res_blocks=int(g["arg_num_res_blocks"]), attention_resolutions=str(g["arg_attention_resolutions"])))
    model, diffusion = create_model_and_diffusion(**a)
    model.load_state_dict(sd, strict=True)
    return model.to(dev).eval(), a


@pytest.mark.parametrize("name", ["tiny32", "mid64", "deep256"])
def test_unet_forward_matches_reference_golden(name):
    g, ks, sd, x, xc, t, y = load_unet_case(name)
    model, _ = build_model(g, sd)
    with torch.no_grad():
        out = model(x.to(dev), t.to(dev), xc.to(dev), y=y.to(dev)).cpu()
    s = int(g["stride"])
    err = (out[:, :, ::s, ::s] - torch.from_numpy(g["out"])).abs().max()
    assert err < 1e-4, float(err)                            # outputs O(0.5) after ~60 stacked fp32 convs
    assert abs(float(out.double().abs().sum()) - g["out_ck"][1]) / g["out_ck"][1] < 1e-5
    # a second call reuses the packed weights/workspace and is deterministic
    with torch.no_grad():
        out2 = model(x.to(dev), t.to(dev), xc.to(dev), y=y.to(dev)).cpu()
    assert torch.equal(out, out2)


def test_fresh_model_outputs_zero_like_reference():
    """zero_module'd output conv => a freshly constructed model predicts exactly 0 (SURVEY 8(c) rule 1)."""
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults
    a = model_and_diffusion_defaults()
    a.update(dict(in_channels=27, out_channels=27, class_cond=True, image_size=32, num_channels=32, num_res_blocks=1))
    m, _ = create_model_and_diffusion(**a)
    m = m.to(dev)
    x = torch.randn(1, 27, 32, 32, device=dev)
    out = m(x, torch.tensor([5.0], device=dev), x, y=torch.tensor([1], device=dev))
    assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize("tag,spec", [("full", ""), ("ddim50", "ddim50"), ("r250", "250")])
@pytest.mark.parametrize("clip", [True, False])
def test_sampler_steps_match_reference(tag, spec, clip):
    """p_sample / ddim_sample / p_mean_variance with a stub model: fused HIP update vs the reference's outputs."""
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    g = np.load(os.path.join(GOLDEN, "diffusion_steps.npz"))
    gen = torch.Generator().manual_seed(7)
    x = torch.randn((3, 27, 8, 8), generator=gen)
    xc = torch.randn((3, 27, 8, 8), generator=gen) * 0.5
    noise = torch.randn((3, 27, 8, 8), generator=gen)
    y = torch.tensor([0, 3, 1])
    d = create_gaussian_diffusion(steps=1000, timestep_respacing=spec)
    c = int(clip)
    t = torch.from_numpy(g[f"step_{tag}_{c}_t"]).long()

    def model(xx, tt, xcond, y=None):
        return _stub(xx.cpu(), tt.cpu(), xcond.cpu(), y.cpu()).to(dev)   # the stub is a test prop, not the product

    orig = torch.randn_like
    torch.randn_like = lambda ref: noise.to(ref.device)
    try:
        ps = d.p_sample(model, x.to(dev), xc.to(dev), t.to(dev), clip_denoised=clip, model_kwargs={"y": y.to(dev)})
        dd = d.ddim_sample(model, x.to(dev), t.to(dev), x_cond=xc.to(dev), clip_denoised=clip, model_kwargs={"y": y.to(dev)})
        de = d.ddim_sample(model, x.to(dev), t.to(dev), x_cond=xc.to(dev), clip_denoised=clip, model_kwargs={"y": y.to(dev)},
                           eta=0.7)
        pm = d.p_mean_variance(model, x.to(dev), t.to(dev), x_cond=xc.to(dev), clip_denoised=clip, model_kwargs={"y": y.to(dev)})
    finally:
        torch.randn_like = orig
    # The fused update follows the reference's fp32 op order with IEEE-exact +,-,*,/ (no FMA contraction);
    # it reproduces a strict-IEEE numpy emulation bit for bit.  The golden comes from PyTorch-CPU on the
    # build host, whose per-timestep scalars (host sqrt/exp of the schedule) differ by one ulp between
    # CPU ISAs - so the pin is "within 2 ulp", not bit-equality.
    def near(a, b):
        return ((a.cpu() - b).abs() <= 3e-7 * b.abs() + 2e-7).all()

    G = lambda k: torch.from_numpy(g[f"step_{tag}_{c}_{k}"])  # noqa: E731
    assert near(ps["pred_xstart"], G("p_x0"))
    assert near(pm["mean"], G("mean"))
    assert near(dd["sample"], G("ddim_sample"))
    assert near(de["sample"], G("ddim_eta_sample"))
    assert near(ps["sample"], G("p_sample"))
    assert near(pm["log_variance"][:, 0, 0, 0], G("logvar"))
    assert near(pm["variance"][:, 0, 0, 0], G("var"))


@pytest.mark.parametrize("tag,spec,ddim", [("ddim10", "ddim10", True), ("p8", "8", False)])
def test_full_sampling_loops_match_reference(tag, spec, ddim):
    """ddim_sample_loop / p_sample_loop on the tiny UNet with the reference's injected noise stream."""
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults
    gl = np.load(os.path.join(GOLDEN, "diffusion_loops.npz"))
    g, ks, sd, _, xc, _, _ = load_unet_case("tiny32")
    a = model_and_diffusion_defaults()
    a.update(dict(in_channels=27, out_channels=27, class_cond=True, num_heads=4, rescale_timesteps=False, image_size=32,
                  num_channels=32, num_res_blocks=1, attention_resolutions="16,8", timestep_respacing=spec))
    model, diffusion = create_model_and_diffusion(**a)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    n = {"i": 0}

    def draw(shape):
        gg = torch.Generator().manual_seed(7000 + n["i"])
        n["i"] += 1
        return torch.randn(tuple(shape), generator=gg)

    x_T = draw((2, 27, 32, 32)).to(dev)
    orig = torch.randn_like
    torch.randn_like = lambda ref: draw(ref.shape).to(ref.device)
    try:
        fn = diffusion.ddim_sample_loop if ddim else diffusion.p_sample_loop
        out = fn(model, (2, 27, 32, 32), x_cond=xc.to(dev), noise=x_T, clip_denoised=True,
                 model_kwargs={"y": torch.tensor([1, 2], device=dev)})
    finally:
        torch.randn_like = orig
    assert n["i"] == int(gl[f"{tag}_ndraws"])               # same RNG call pattern as the reference
    err = (out.cpu() - torch.from_numpy(gl[f"{tag}_sample"])).abs().max()
    assert err < 1e-4, float(err)                           # 8-10 recurrent UNet evaluations (50 DDIM steps measure 1.9e-5, test_e2e_gpu.py)


def test_c_abi_error_convention():
    """Bad arguments come back as negative status + message (never an exception across the ABI, never a crash);
    the Python mirrors re-raise with the reference's exception types."""
    import ctypes as C
    from humanliff_amd import _lib
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults
    L = _lib.lib()
    x = torch.zeros((1, 8, 8, 24), device=dev)          # Cin not a multiple of 16
    w = torch.zeros((32, 24, 3, 3), device=dev)
    out = torch.zeros((1, 8, 8, 32), device=dev)
    scratch = torch.zeros(1 << 20, device=dev)
    rc = L.hl_conv2d_nhwc(_lib.ptr(x), 1, 8, 8, 24, _lib.ptr(w), None, 32, 3, 1, 0, None, None, 0, None, _lib.ptr(out),
                          _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr())
    assert rc == -1 and b"multiple of 16" in L.hl_last_error()
    rc = L.hl_diffusion_step(7, _lib.ptr(out), _lib.ptr(out), None, _lib.ptr(out), _lib.ptr(out), _lib.ptr(out), None, 16, 1, 10, 1,
                             None, _lib.stream_ptr())
    assert rc == -1 and b"mode" in L.hl_last_error()
    rc = L.hl_render_importance(_lib.ptr(out), _lib.ptr(out), _lib.ptr(out), _lib.ptr(out), None, _lib.ptr(out), 4, 1024, 1024,
                                _lib.ptr(out), _lib.stream_ptr())
    assert rc == -2                                       # unsupported size, HL_ERR_UNSUPPORTED
    # state_dict with a missing / mis-shaped tensor is reported by name
    a = model_and_diffusion_defaults()
    a.update(dict(in_channels=27, out_channels=27, class_cond=True, image_size=32, num_channels=32, num_res_blocks=1))
    m, _ = create_model_and_diffusion(**a)
    m = m.to(dev).eval()
    cfg = m._cfg()
    sd = {k: v for k, v in m.state_dict().items() if k != "middle_block.1.qkv.weight"}
    n = len(sd)
    names = (C.c_char_p * n)(*[k.encode() for k in sd])
    ptrs = (C.c_void_p * n)(*[v.data_ptr() for v in sd.values()])
    numels = (C.c_int64 * n)(*[v.numel() for v in sd.values()])
    packed = torch.empty(L.hl_unet_packed_bytes(C.byref(cfg)) // 4 + 64, device=dev)
    h = C.c_void_p()
    rc = L.hl_unet_create(C.byref(cfg), n, names, ptrs, numels, _lib.ptr(packed), _lib.stream_ptr(), C.byref(h))
    assert rc == -1 and b"middle_block.1.qkv.weight" in L.hl_last_error()
    # Python mirror: shape assertions like the reference (unet.py:585)
    with pytest.raises(AssertionError):
        m(torch.zeros(2, 27, 32, 32, device=dev), torch.tensor([1, 2], device=dev), torch.zeros(2, 27, 32, 32, device=dev),
          y=torch.tensor([0], device=dev))
    with pytest.raises(_lib.HipCallError):                # H, W not divisible by the total downsampling
        m(torch.zeros(1, 27, 36, 36, device=dev), torch.tensor([1], device=dev), torch.zeros(1, 27, 36, 36, device=dev),
          y=torch.tensor([0], device=dev))


def test_training_path_matches_inference_forward_and_twin():
    """The three statements of the network agree on the GPU: the fused inference forward, the differentiable HIP path (unet_train.py)
    and the PyTorch-op twin (unet_autograd.py, MIOpen here)."""
    from tests.test_train_loss_cpu import inputs, tiny_model
    from humanliff_amd.improved_diffusion.unet_train import forward_train
    model, diffusion = tiny_model()
    model = model.to(dev)
    x0, xc = (t.to(dev) for t in inputs())
    t, y = torch.tensor([999, 17], device=dev), torch.tensor([3, 0], device=dev)
    with torch.no_grad():
        hip = model(x0, t, xc, y=y)
        train = forward_train(model, x0, t, xc, y)
        twin = forward_autograd(model, x0, t, xc, y=y)
    scale = max(1.0, float(twin.abs().max()))
    assert (hip - train).abs().max() < 5e-5 * scale            # same kernels, different fusion (materialised GroupNorm, no concat buffers)
    assert (hip - twin).abs().max() < 2e-4 * scale             # MIOpen vs the HIP kernels: different summation orders


@pytest.mark.parametrize("N,C,H,W,Cout,ks,stride,ups,mode,with_gn,expect_stats", [
    (2, 64, 256, 128, 192, 3, 1, 0, 0, True, True),     # Winograd F(4x4) after the k_gn_apply pass, residual; slot = (32x16 block, round, wave)
    (2, 64, 256, 128, 192, 3, 1, 0, 3, True, True),     # Winograd F(2x2) (HL_CONV_FP32_F23); slot = (16x8 block, parity)
    (4, 96, 64, 64, 192, 3, 1, 1, 0, False, True),      # nearest x2 + 3x3 (Upsample) through the F(4x4) kernel
    (4, 384, 32, 32, 384, 3, 1, 0, 0, True, True),      # Winograd over 3 input-channel slabs: statistics from k_splitk_finish_st
    (2, 96, 32, 64, 192, 3, 1, 1, 0, False, True),      # nearest x2 + 3x3 (Upsample) through the Winograd kernel
    (2, 192, 64, 64, 384, 1, 1, 0, 2, False, True),     # 1x1 skip / zero-conv on the direct kernel: slot = a wave's 32 rows
    (2, 96, 32, 32, 96, 3, 2, 0, 2, False, True),       # stride 2 (Downsample)
    (4, 768, 8, 8, 768, 3, 1, 0, 2, True, True),        # 8x8 level: 16 slabs, 64 pixels per image = 2 slots
    (1, 96, 160, 96, 96, 3, 1, 0, 2, False, True),      # ragged last pixel tile of the 256-row tiling
    (2, 64, 16, 16, 64, 3, 1, 0, 2, False, True),       # register-staged small-tile kernel over K slabs: statistics from k_splitk_finish_st
    (2, 96, 12, 12, 96, 3, 1, 0, 2, False, False),      # 144 pixels per image: 32-row slots would straddle images
])
def test_conv_epilogue_groupnorm_statistics(N, C, H, W, Cout, ks, stride, ups, mode, with_gn, expect_stats):
    """The kernel that stores a tensor also emits the GroupNorm statistics the next layer needs (per slot of pixels and channel
    (sum, sumsq), folded in a fixed order): the affine from them must equal GroupNorm32 of the stored tensor (nn.py:17-19,100)."""
    from humanliff_amd import _lib
    L = _lib.lib()
    if Cout % 32:
        pytest.skip("GroupNorm32 needs C % 32 == 0")
    g = torch.Generator().manual_seed(N * 1000 + C + H + Cout)
    x = torch.randn((N, C, H, W), generator=g) * 1.3 + 0.2
    w = torch.randn((Cout, C, ks, ks), generator=g) / (C * ks * ks) ** 0.5
    b = torch.randn((Cout,), generator=g)
    gamma, beta = torch.randn(Cout, generator=g) * 0.2 + 1, torch.randn(Cout, generator=g) * 0.2
    cA = cB = res = None
    Hv, Wv = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = (Hv + 2 * (ks // 2) - ks) // stride + 1, (Wv + 2 * (ks // 2) - ks) // stride + 1
    if with_gn:
        cA, cB = torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g) * 0.3
        res = torch.randn((N, Cout, Ho, Wo), generator=g)
    d = lambda t: None if t is None else t.contiguous().to(dev)  # noqa: E731
    xin, wd, bd, cAd, cBd, gd_, be_ = d(nhwc(x)), d(w), d(b), d(cA), d(cB), d(gamma), d(beta)
    rd = d(nhwc(res)) if res is not None else None
    out = torch.empty((N, Ho, Wo, Cout), device=dev)
    nA, nB = torch.empty((N, Cout), device=dev), torch.empty((N, Cout), device=dev)
    scratch = torch.empty(((Cout + 63) // 64 * 64) * C * ks * ks * 5 + 256 + (8 << 20) + N * C * H * W + N * Ho * Wo * Cout // 8 + N * 8192,
                          device=dev)
    import ctypes
    used = ctypes.c_int(-1)
    _lib.check(L.hl_conv2d_nhwc_gn(mode, _lib.ptr(xin), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Cout, ks, stride, ups, _lib.ptr(cAd),
                                   _lib.ptr(cBd), 1 if with_gn else 0, _lib.ptr(rd), _lib.ptr(out), _lib.ptr(gd_), _lib.ptr(be_),
                                   _lib.ptr(nA), _lib.ptr(nB), ctypes.byref(used), _lib.ptr(scratch), scratch.numel() * 4,
                                   _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert (used.value > 0) == expect_stats, used.value
    # the convolution itself (the statistics ride its epilogue: nothing about the stored values may change)
    ref = hip_conv(x, w, b, ks, stride, ups, cA=cA, cB=cB, silu=1 if with_gn else 0, res=res, mode=mode)
    y = nchw(out.cpu())
    assert torch.equal(y, ref)
    # GroupNorm32 of the stored tensor, in float64
    yd = y.double().reshape(N, 32, -1)
    mean, var = yd.mean(dim=2), yd.var(dim=2, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    cg = Cout // 32
    wantA = (rstd[:, :, None] * gamma.double().reshape(1, 32, cg)).reshape(N, Cout)
    wantB = beta.double()[None] - (mean[:, :, None].expand(N, 32, cg).reshape(N, Cout)) * wantA
    assert (nA.cpu().double() - wantA).abs().max() < 2e-6 * wantA.abs().max()
    assert (nB.cpu().double() - wantB).abs().max() < 2e-6 * max(1.0, float(wantB.abs().max()))


@pytest.mark.parametrize("N,H,W,C,Cout,res,gn", [(4, 64, 64, 384, 192, True, False), (2, 128, 128, 192, 192, False, False), (1, 256, 256, 576, 192, True, False),
                                               (4, 32, 32, 384, 1152, False, True), (2, 64, 64, 768, 384, True, False)])
def test_conv1x1_fp16x2_products_match_float64(N, H, W, C, Cout, res, gn):
    """The 1x1 convolutions of the DEFAULT mode run on k_conv1_h2s (128-pixel tiles, two workgroups per CU) from 12 workgroups' worth of 256 pixels x 192 channels on: two fp16 planes per operand, three partial products, fp32
    accumulation.  Against the float64 convolution of the fp32 operands: the error of an fp32 convolution's class (bound 4e-6 of the output scale;
    a plain fp32 MFMA kernel measures ~1e-6 here), with residual and with the GroupNorm pre-pass of the attention's qkv convolution."""
    from humanliff_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(N + C + Cout)
    x = torch.randn((N, H, W, C), generator=g) * 1.5
    w = torch.randn((Cout, C, 1, 1), generator=g) / C ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn((N, H, W, Cout), generator=g)
    cA, cB = torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g) * 0.1
    xin = (x * cA[:, None, None, :] + cB[:, None, None, :]).to(dev).cpu() if gn else x     # (the pre-pass runs in fp32 on the GPU)
    ref = torch.einsum("nhwc,oc->nhwo", xin.double(), w[:, :, 0, 0].double()) + b.double()
    if res:
        ref = ref + r.double()
    xd, wd, bd, rd, ad, bd2 = (t.to(dev) for t in (x, w, b, r, cA, cB))
    out = torch.zeros((N, H, W, Cout), device=dev)
    scratch = torch.empty(Cout * C * 8 + 256 + (8 << 20) + N * H * W * C, device=dev)
    import ctypes
    plan = ctypes.c_int(-1)
    _lib.check(L.hl_conv2d_nhwc_mode(_lib.HL_CONV_FP32, _lib.ptr(xd), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Cout, 1, 1, 0, _lib.ptr(ad) if gn else None,
                                     _lib.ptr(bd2) if gn else None, 0, _lib.ptr(rd) if res else None, _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4,
                                     _lib.stream_ptr()), "hl_conv2d_nhwc_mode")
    # the fp32 direct kernel on the same operands, for scale
    out32 = torch.zeros_like(out)
    _lib.check(L.hl_conv2d_nhwc_mode(_lib.HL_CONV_FP32_DIRECT, _lib.ptr(xd), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Cout, 1, 1, 0, _lib.ptr(ad) if gn else None,
                                     _lib.ptr(bd2) if gn else None, 0, _lib.ptr(rd) if res else None, _lib.ptr(out32), _lib.ptr(scratch), scratch.numel() * 4,
                                     _lib.stream_ptr()), "hl_conv2d_nhwc_mode")
    scale = float(ref.abs().mean())
    e2, e32 = float((out.cpu().double() - ref).abs().max()), float((out32.cpu().double() - ref).abs().max())
    print(f"1x1 {C}->{Cout} @{H}x{W} N{N}: fp16x2 max-abs {e2:.2e}, fp32 direct {e32:.2e} (output mean-abs {scale:.2f})")
    assert not torch.equal(out, out32)                      # the default mode really took the other kernel
    l2, l32 = float((out.cpu().double() - ref).norm() / ref.norm()), float((out32.cpu().double() - ref).norm() / ref.norm())
    print(f"    rel-L2: fp16x2 {l2:.2e}, fp32 direct {l32:.2e}")
    # measured on MI355X: max-abs 3.6e-6 ... 7.5e-6 against 2.9e-6 ... 9.3e-6 of the fp32 direct kernel; rel-L2 3.1e-7 ... 3.7e-7
    assert e2 < 6e-6 * max(1.0, scale) * (C / 384) ** 0.5, (e2, scale)
    assert l2 < 8e-7 and l2 < 4 * l32 + 2e-7, (l2, l32)


@pytest.mark.parametrize("N,H,W,C,Cout,res,gn,ups", [(1, 256, 256, 192, 192, True, True, 0), (4, 128, 128, 192, 192, False, False, 0), (4, 64, 64, 384, 384, True, True, 0),
                                                   (4, 32, 32, 384, 384, False, False, 1), (2, 64, 64, 768, 384, False, True, 0), (3, 112, 144, 96, 192, True, False, 0)])
def test_conv3x3_fp16x2_products_match_float64(N, H, W, C, Cout, res, gn, ups):
    """The 3x3 / stride-1 layers of the DEFAULT mode from 100 workgroups' worth of work on (256 pixels x 192 channels each) run on k_conv_h2s: a direct
    convolution on 8x16-pixel tiles, two fp16 planes per operand, three partial products, fp32 accumulation - against the float64 convolution of the fp32
    operands, with the GroupNorm pre-pass writing the two-plane image, the residual, the nearest-x2 upsample in the patch gather, and a non-square image
    whose tile rows end in the middle of the last 16-row band."""
    import torch.nn.functional as F
    from humanliff_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(N + C + Cout + H)
    x = torch.randn((N, H, W, C), generator=g) * 1.5
    w = torch.randn((Cout, C, 3, 3), generator=g) / (9 * C) ** 0.5
    b = torch.randn(Cout, generator=g)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    r = torch.randn((N, Ho, Wo, Cout), generator=g)
    cA, cB = torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g) * 0.1
    xin = F.silu((x * cA[:, None, None, :] + cB[:, None, None, :]).to(dev)).cpu() if gn else x     # (the pre-pass runs in fp32 on the GPU)
    xi = xin.permute(0, 3, 1, 2).double()
    if ups:
        xi = F.interpolate(xi, scale_factor=2, mode="nearest")
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = F.conv2d(xi, w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    if res:
        ref = ref + r.double()
    xd, wd, bd, rd, ad, bd2 = (t.to(dev) for t in (x, w, b, r, cA, cB))
    scratch = torch.empty(Cout * C * 9 * 8 + 256 + (64 << 20) + N * H * W * C, device=dev)
    outs = {}
    for mode in (_lib.HL_CONV_FP32, _lib.HL_CONV_FP32_DIRECT):
        out = torch.zeros((N, Ho, Wo, Cout), device=dev)
        _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(xd), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Cout, 3, 1, ups, _lib.ptr(ad) if gn else None,
                                         _lib.ptr(bd2) if gn else None, 1 if gn else 0, _lib.ptr(rd) if res else None, _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4,
                                         _lib.stream_ptr()), "hl_conv2d_nhwc_mode")
        outs[mode] = out.cpu().double()
    out, out32 = outs[_lib.HL_CONV_FP32], outs[_lib.HL_CONV_FP32_DIRECT]
    scale = float(ref.abs().mean())
    e2, e32 = float((out - ref).abs().max()), float((out32 - ref).abs().max())
    l2, l32 = float((out - ref).norm() / ref.norm()), float((out32 - ref).norm() / ref.norm())
    print(f"3x3 {C}->{Cout} @{Ho}x{Wo} N{N}: fp16x2 max-abs {e2:.2e} rel-L2 {l2:.2e}; fp32 direct {e32:.2e} / {l32:.2e} (output mean-abs {scale:.2f})")
    assert not torch.equal(out, out32)                      # the default mode really took another kernel
    # measured on MI355X: max-abs 6.8e-6 ... 1.9e-5 (fp32 direct kernel 7.5e-6 ... 1.5e-5), rel-L2 5.0e-7 ... 1.2e-6 (fp32 direct 4.2e-7 ... 6.3e-7)
    assert e2 < 6e-6 * max(1.0, scale) * (9 * C / 384) ** 0.5, (e2, scale)     # (the 1x1 test's bound at this reduction length)
    assert l2 < 1.6e-6 and l2 < 4 * l32 + 2e-7, (l2, l32)


@pytest.mark.parametrize("N,H,W,C,Cout", [(1, 256, 256, 192, 192), (2, 128, 128, 96, 192), (4, 64, 64, 384, 384), (3, 96, 160, 64, 192)])
def test_conv3x3_stride2_fp16x2_products_match_float64(N, H, W, C, Cout):
    """The Downsample convolutions (3x3, stride 2, unet.py:100) of the DEFAULT mode from 32 workgroups' worth of output on run on k_conv_h2d: the input patch staged
    de-interleaved by row / column parity, fp16x2 products, fp32 accumulation - against the float64 convolution of the fp32 operands and the fp32 direct kernel."""
    import torch.nn.functional as F
    from humanliff_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(N + C + Cout + H)
    x = torch.randn((N, H, W, C), generator=g) * 1.5
    w = torch.randn((Cout, C, 3, 3), generator=g) / (9 * C) ** 0.5
    b = torch.randn(Cout, generator=g)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=2, padding=1).permute(0, 2, 3, 1)
    xd, wd, bd = (t.to(dev) for t in (x, w, b))
    scratch = torch.empty(Cout * C * 9 * 8 + 256 + (64 << 20), device=dev)
    outs = {}
    for mode in (_lib.HL_CONV_FP32, _lib.HL_CONV_FP32_DIRECT):
        out = torch.zeros((N, H // 2, W // 2, Cout), device=dev)
        _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(xd), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Cout, 3, 2, 0, None, None, 0, None, _lib.ptr(out), _lib.ptr(scratch),
                                         scratch.numel() * 4, _lib.stream_ptr()), "hl_conv2d_nhwc_mode")
        outs[mode] = out.cpu().double()
    out, out32 = outs[_lib.HL_CONV_FP32], outs[_lib.HL_CONV_FP32_DIRECT]
    scale = float(ref.abs().mean())
    e2, e32 = float((out - ref).abs().max()), float((out32 - ref).abs().max())
    l2, l32 = float((out - ref).norm() / ref.norm()), float((out32 - ref).norm() / ref.norm())
    print(f"3x3 stride 2 {C}->{Cout} @{H // 2}x{W // 2} N{N}: fp16x2 max-abs {e2:.2e} rel-L2 {l2:.2e}; fp32 direct {e32:.2e} / {l32:.2e} (output mean-abs {scale:.2f})")
    assert not torch.equal(out, out32)                      # the default mode really took another kernel
    assert e2 < 6e-6 * max(1.0, scale) * (9 * C / 384) ** 0.5, (e2, scale)
    assert l2 < 1.6e-6 and l2 < 4 * l32 + 2e-7, (l2, l32)


_SCALE_CASES = [(0, 0), (-6, 0), (-12, 0), (-18, 0), (0, -10), (0, -5), (0, 5), (0, 10), (-12, -10), (-18, 10), (6, -10)]


@pytest.mark.parametrize("kind,N,H,W,C,Cout,gn", [("3x3", 4, 64, 64, 384, 384, False), ("3x3", 1, 128, 128, 192, 192, True), ("1x1", 4, 64, 64, 384, 192, False),
                                                  ("1x1", 2, 64, 64, 384, 1152, True), ("s2", 2, 128, 128, 96, 192, False), ("3x3k", 4, 32, 32, 384, 384, False)])
def test_conv_fp16x2_products_are_scale_invariant(kind, N, H, W, C, Cout, gn):
    """Round 6: the split products of the DEFAULT mode must not depend on the magnitude of their operands.  fp16 has a 5-bit
    exponent; the weight planes are scaled per output channel by a power of two at pack time (k_wscale_h2) and a raw input by the power of two that its
    producers' sum x^2 bounds (act_scale_totals; the single-op entry point forms the totals itself), both undone exactly in the epilogue.  One float64
    reference per shape: scaling the weights by 2^a and the input by 2^b scales it (and the fp32 direct kernel's result) exactly, so every (a, b) is
    judged against the SAME numbers: rel-L2 of the fp16x2 kernel <= 1.3 x the fp32 direct kernel's + 5e-8 at every scale - the unscaled planes of
    round 5 gave 4.6e-5 at 2^-6 and 2.9e-3 at 2^-12 on the weights, 2e-5 at 2^-10 on the activations.  With a GroupNorm given as coefficient ARRAYS (gn) nothing
    is known about the normalised tensor's magnitude, so only the weights are scaled there; '3x3k' is the split-K dispatch (raw slabs scaled before they
    are summed)."""
    from humanliff_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(N + C + Cout + H)
    ks, stride = (1, 1) if kind == "1x1" else (3, 2 if kind == "s2" else 1)
    x = torch.randn((N, H, W, C), generator=g) * 1.5
    w = torch.randn((Cout, C, ks, ks), generator=g) / (ks * ks * C) ** 0.5
    w = w * torch.exp2(torch.randint(-6, 1, (Cout, 1, 1, 1), generator=g).float())        # channels of different magnitude inside one layer
    b = torch.randn(Cout, generator=g) * 0.1
    cA, cB = torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g) * 0.1
    act = 1 if (gn and ks == 3) else 0
    xin = x * cA[:, None, None, :] + cB[:, None, None, :] if gn else x
    if act:
        xin = F.silu(xin.to(dev)).cpu()
    elif gn:
        xin = xin.to(dev).cpu()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = F.conv2d(xin.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=stride, padding=ks // 2).permute(0, 2, 3, 1)
    Ho, Wo = ref.shape[1], ref.shape[2]
    scratch = torch.empty(Cout * C * ks * ks * 8 + 256 + (64 << 20) + N * H * W * C, device=dev)
    rows = []
    for a, bx in _SCALE_CASES:
        if gn and bx != 0:
            continue
        sw, sx = 2.0 ** a, 2.0 ** bx
        xd, wd, bd = (x * sx).to(dev), (w * sw).to(dev), (b * (sw * sx)).to(dev)
        ad, bd2 = cA.to(dev), (cB * 1.0).to(dev)
        outs = {}
        for mode in (_lib.HL_CONV_FP32, _lib.HL_CONV_FP32_DIRECT):
            out = torch.zeros((N, Ho, Wo, Cout), device=dev)
            _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(xd), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Cout, ks, stride, 0, _lib.ptr(ad) if gn else None,
                                             _lib.ptr(bd2) if gn else None, act, None, _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()),
                       "hl_conv2d_nhwc_mode")
            outs[mode] = out.cpu().double() / (sw * sx)
        out, out32 = outs[_lib.HL_CONV_FP32], outs[_lib.HL_CONV_FP32_DIRECT]
        assert not torch.equal(out, out32)                  # the default mode really took the fp16x2 kernel
        l2, l32 = float((out - ref).norm() / ref.norm()), float((out32 - ref).norm() / ref.norm())
        rows.append((a, bx, l2, l32))
    print(f"{kind} {C}->{Cout} @{Ho}x{Wo} N{N} gn={gn}: " + "; ".join(f"w 2^{a} x 2^{bx}: {l2:.2e} (fp32 {l32:.2e})" for a, bx, l2, l32 in rows))
    for a, bx, l2, l32 in rows:
        assert l2 <= 1.3 * l32 + 5e-8, (a, bx, l2, l32)


@pytest.mark.parametrize("kind,N,H,W,C,Cout", [("1x1", 4, 64, 64, 384, 192), ("3x3", 4, 64, 64, 384, 384), ("s2", 2, 128, 128, 96, 192)])
def test_conv_fp16x2_scale_from_group_totals(kind, N, H, W, C, Cout):
    """Inside hl_unet_forward the power-of-two scale of a raw input comes from the fixed-point group totals its producers left (|x| <= sqrt(sum x^2),
    act_scale_totals); the single-op entry points use an exact abs-max pass instead.  hl_debug_set_single_op_scale_source(1) makes them form the totals (one pass,
    k_tensor_totals) and hand those to the kernels: the network's scale source on single layers.  Same float64 reference at every scale; the totals resolve sum x^2
    down to ~1e-7 per contribution and poison above 2.7e11 per group, so the window is narrower than the abs-max's: inside it (x 2^-8 ... 2^6 at these sizes) the
    same bits as at scale 1, outside it the kernels fall back to scale 1 (round 5's behaviour) and only the looser bound holds."""
    from humanliff_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(N + C + Cout + H + 1)
    ks, stride = (1, 1) if kind == "1x1" else (3, 2 if kind == "s2" else 1)
    x = torch.randn((N, H, W, C), generator=g) * 1.5
    w = torch.randn((Cout, C, ks, ks), generator=g) / (ks * ks * C) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=stride, padding=ks // 2).permute(0, 2, 3, 1)
    Ho, Wo = ref.shape[1], ref.shape[2]
    scratch = torch.empty(Cout * C * ks * ks * 8 + 256 + (64 << 20), device=dev)
    wd = w.to(dev)
    rows = {}
    try:
        for src in (0, 1):
            _lib.check(L.hl_debug_set_single_op_scale_source(src))
            for bx in (-14, -8, -4, 0, 6, 12):
                sx = 2.0 ** bx
                xd, bd = (x * sx).to(dev), (b * sx).to(dev)
                out = torch.zeros((N, Ho, Wo, Cout), device=dev)
                _lib.check(L.hl_conv2d_nhwc_mode(_lib.HL_CONV_FP32, _lib.ptr(xd), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Cout, ks, stride, 0, None, None, 0, None,
                                                 _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()), "hl_conv2d_nhwc_mode")
                o = out.cpu().double() / sx
                rows[(src, bx)] = (float((o - ref).norm() / ref.norm()), o)
    finally:
        _lib.check(L.hl_debug_set_single_op_scale_source(0))
    print(f"{kind} {C}->{Cout}: " + "; ".join(f"{'totals' if s_ else 'absmax'} x 2^{bx}: {v[0]:.2e}" for (s_, bx), v in rows.items()))
    base = rows[(0, 0)][0]
    for bx in (-14, -8, -4, 0, 6, 12):
        assert rows[(0, bx)][0] <= base * 1.001 + 1e-9, (bx, rows[(0, bx)][0], base)           # abs-max: exact at every magnitude
    for bx in (-8, -4, 0, 6):
        assert torch.equal(rows[(1, bx)][1], rows[(1, 0)][1]), bx                                # totals: the same bits inside their window ...
        assert rows[(1, bx)][0] <= base * 1.05 + 2e-8, (bx, rows[(1, bx)][0], base)              # ... and the abs-max figure (the two scales may differ by a power of two)
    for bx in (-14, 12):
        assert rows[(1, bx)][0] < 5e-5, (bx, rows[(1, bx)][0])                                    # outside it: scale 1, the unscaled planes' error at that magnitude


def test_conv_fp16x2_activations_beyond_fp16_range_match_fp32():
    """|x| = 7e4 is beyond fp16 (65504): round 5's planes saturated silently (v_cvt_pkrtz never produces inf).  The raw input is now scaled by the power of two that
    sqrt(sum x^2) bounds, so the same launch matches the float64 reference like the fp32 direct kernel does."""
    from humanliff_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(77)
    N, H, W, C, Cout = 2, 64, 64, 384, 192
    x = torch.randn((N, H, W, C), generator=g) * 1.5
    x[0, 3, 5, 7] = 7.0e4; x[1, 60, 2, 300] = -9.0e4; x[0, 0, 0, 0] = 6.6e4
    w = torch.randn((Cout, C, 1, 1), generator=g) / C ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = torch.einsum("nhwc,oc->nhwo", x.double(), w[:, :, 0, 0].double()) + b.double()
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    scratch = torch.empty(Cout * C * 8 + 256 + (16 << 20), device=dev)
    outs = {}
    for mode in (_lib.HL_CONV_FP32, _lib.HL_CONV_FP32_DIRECT):
        out = torch.zeros((N, H, W, Cout), device=dev)
        _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(xd), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Cout, 1, 1, 0, None, None, 0, None, _lib.ptr(out), _lib.ptr(scratch),
                                         scratch.numel() * 4, _lib.stream_ptr()), "hl_conv2d_nhwc_mode")
        outs[mode] = out.cpu().double()
    out, out32 = outs[_lib.HL_CONV_FP32], outs[_lib.HL_CONV_FP32_DIRECT]
    assert not torch.equal(out, out32) and bool(torch.isfinite(out).all())
    l2, l32 = float((out - ref).norm() / ref.norm()), float((out32 - ref).norm() / ref.norm())
    big = float((out[0, 3, 5] - ref[0, 3, 5]).abs().max() / ref[0, 3, 5].abs().max())
    print(f"|x| up to 9e4: fp16x2 rel-L2 {l2:.2e}, fp32 direct {l32:.2e}; pixel with 7e4: rel max {big:.2e}")
    assert l2 <= 1.3 * l32 + 5e-8 and big < 2e-6, (l2, l32, big)
