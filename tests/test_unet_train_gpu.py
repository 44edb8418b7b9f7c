"""HIP backward of the UNet: the backward kernels op by op against PyTorch's autograd of the same op, and
GaussianDiffusion.training_losses(...).backward() end to end against the REFERENCE's loss and parameter gradients
(tests/golden/train_loss_tiny32.npz <- tests/golden/gen_golden_train_loss.py).  Everything goes through the C ABI."""
import os

import numpy as np
import pytest
import torch
from tests.unet_autograd_twin import forward_autograd
import torch.nn.functional as F

from tests.golden_util import GOLDEN

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("N,Cin,H,W,Cout,ks,stride,ups", [
    (2, 64, 32, 32, 64, 3, 1, 0),       # one 64x64 block per tap
    (2, 192, 64, 64, 192, 3, 1, 0),     # production channel count: 3 x 3 blocks x 9 taps, K slabs
    (1, 96, 40, 24, 160, 3, 1, 0),      # ragged channel blocks (96 = 64 + 32, 160 = 128 + 32), non-square
    (2, 64, 32, 32, 64, 3, 2, 0),       # Downsample: stride 2
    (2, 64, 16, 16, 96, 3, 1, 1),       # Upsample: conv on the nearest-x2 image
    (2, 128, 16, 16, 384, 1, 1, 0),     # 1x1 (qkv / skip / zero-conv)
    (1, 32, 32, 32, 27, 3, 1, 0),       # the 27-channel output conv (gradient padded to 28 channels by the caller)
    (3, 27, 16, 16, 32, 3, 1, 0),       # the 27-channel input conv (input padded to 32 channels by the caller)
    (2, 32, 4, 4, 32, 3, 1, 0),         # 16 pixels per image: slabs and pixel pairs cross images
])
def test_conv_backward_matches_torch_autograd(N, Cin, H, W, Cout, ks, stride, ups):
    """_Conv: d input through the forward conv kernels on flipped weights, d weight / d bias by k_conv_wgrad_t (3x3) / k_conv_wgrad (1x1)."""
    from humanliff_amd.improved_diffusion import unet_train as ut
    g = torch.Generator().manual_seed(N * 100 + Cin + Cout + H)
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, ks, ks), generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn((Cout,), generator=g)
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest") if ups else xr, wr, br, stride=stride, padding=ks // 2)
    cot = torch.randn(yr.shape, generator=g)
    (yr * cot.double()).sum().backward()
    xd = ut._pad_c(nhwc(x), 16).to(dev).requires_grad_(True)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = ut._Conv.apply(xd, wd, bd, stride, ups)
    assert (nchw(y.detach().cpu()) - F.conv2d(xi, w, b, stride=stride, padding=ks // 2)).abs().max() < 3e-5
    (y * nhwc(cot).to(dev)).sum().backward()
    dx = nchw(xd.grad.cpu())[:, :Cin]
    sx, sw, sb = float(xr.grad.abs().max()), float(wr.grad.abs().max()), float(br.grad.abs().max())
    assert (dx.double() - xr.grad).abs().max() < 2e-5 * max(1.0, sx)
    assert (wd.grad.cpu().double() - wr.grad).abs().max() < 1e-5 * sw          # K up to 8192 pixels per element, fp32 partial sums
    assert (bd.grad.cpu().double() - br.grad).abs().max() < 1e-5 * sb


@pytest.mark.parametrize("ks", [3, 1])
def test_wgrad_is_bit_reproducible_and_checks_its_scratch(ks):
    """hl_conv2d_wgrad_nhwc_ws through the C ABI: the 3x3 and the 1x1 kernel sum their slabs in a fixed order (no atomics) - two runs
    agree bit for bit and match float64; the gradients are stored, not accumulated; a scratch smaller than
    hl_conv2d_wgrad_scratch_bytes is refused."""
    from humanliff_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    N, H, W, Cin, Cout = 2, 48, 40, 128, 192
    x = torch.randn((N, Cin, H, W), generator=g)
    dy = torch.randn((N, Cout, H, W), generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, ks, ks), dy.double(), padding=ks // 2)
    xd, dyd = nhwc(x).to(dev), nhwc(dy).to(dev)
    nbytes = L.hl_conv2d_wgrad_scratch_bytes(N, H, W, Cin, Cout, ks, 1, 0, Cout, Cin)
    assert nbytes > 0
    outs = []
    for _ in range(2):
        dw, db = torch.full((Cout, Cin, ks, ks), 7.0, device=dev), torch.full((Cout,), 7.0, device=dev)    # stored, not accumulated
        part = torch.empty(nbytes // 4, device=dev)
        with _lib.on(dev):
            _lib.check(L.hl_conv2d_wgrad_nhwc_ws(_lib.ptr(xd), N, H, W, Cin, _lib.ptr(dyd), Cout, ks, 1, 0, _lib.ptr(dw), Cout, Cin, _lib.ptr(db),
                                                 _lib.ptr(part), nbytes, _lib.stream_ptr()), "hl_conv2d_wgrad_nhwc_ws")
        outs.append((dw.cpu(), db.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert (outs[0][0].double() - ref).abs().max() < 1e-5 * float(ref.abs().max())
    assert (outs[0][1].double() - dy.double().sum((0, 2, 3))).abs().max() < 1e-5 * float(dy.double().sum((0, 2, 3)).abs().max())
    with _lib.on(dev):
        rc = L.hl_conv2d_wgrad_nhwc_ws(_lib.ptr(xd), N, H, W, Cin, _lib.ptr(dyd), Cout, ks, 1, 0, _lib.ptr(dw), Cout, Cin, _lib.ptr(db),
                                       _lib.ptr(part), nbytes - 4, _lib.stream_ptr())
    assert rc != 0 and b"scratch too small" in L.hl_last_error()


@pytest.mark.parametrize("N,C,H,W,use_ss,silu", [(2, 64, 16, 16, True, True), (2, 192, 32, 32, True, True), (1, 96, 20, 12, False, True),
                                                   (3, 128, 8, 8, False, False), (2, 384, 64, 64, True, True)])
def test_groupnorm_backward_matches_torch_autograd(N, C, H, W, use_ss, silu):
    """_GroupNormAct: GroupNorm32 [* (1 + scale) + shift] [SiLU] forward and every gradient (x, gamma, beta, scale/shift)."""
    from humanliff_amd.improved_diffusion import unet_train as ut
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn((N, C, H, W), generator=g) * 1.5 + 0.3
    gamma, beta = torch.randn(C, generator=g) * 0.2 + 1, torch.randn(C, generator=g) * 0.2
    ss = torch.randn((N, 2 * C), generator=g) * 0.3 if use_ss else None
    cot = torch.randn((N, C, H, W), generator=g)
    leaf = lambda t: None if t is None else t.double().requires_grad_(True)  # noqa: E731
    xr, gr, br, sr = leaf(x), leaf(gamma), leaf(beta), leaf(ss)
    u = F.group_norm(xr, 32, gr, br, eps=1e-5)
    if use_ss:
        u = u * (1 + sr[:, :C, None, None]) + sr[:, C:, None, None]
    yr = u * torch.sigmoid(u) if silu else u
    (yr * cot.double()).sum().backward()
    xd = nhwc(x).to(dev).requires_grad_(True)
    gd_, bd_ = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    sd_ = ss.to(dev).requires_grad_(True) if use_ss else None
    y = ut._GroupNormAct.apply(xd, gd_, bd_, sd_, silu)
    assert (nchw(y.detach().cpu()).double() - yr.detach()).abs().max() < 2e-5
    (y * nhwc(cot).to(dev)).sum().backward()
    rel = lambda a, b: float((a.cpu().double() - b).abs().max() / max(1e-6, float(b.abs().max())))  # noqa: E731
    assert rel(nchw(xd.grad), xr.grad) < 2e-5
    assert rel(gd_.grad, gr.grad) < 2e-5 and rel(bd_.grad, br.grad) < 2e-5
    if use_ss:
        assert rel(sd_.grad, sr.grad) < 2e-5


def test_groupnorm_backward_is_bit_reproducible():
    """k_gn_bwd_reduce / k_gn_bwd_fin add their partial sums in a fixed order (no float atomics): the same bits on every run, also at the
    256 x 256 production level where 256 workgroups per image contribute to every (n, c) sum."""
    from humanliff_amd.improved_diffusion import unet_train as ut
    for N, C, H, W in [(2, 192, 256, 256), (3, 96, 20, 12), (1, 576, 32, 32)]:
        g = torch.Generator().manual_seed(C)
        x = (torch.randn((N, H, W, C), generator=g) * 1.5 + 0.3).to(dev)
        gamma, beta = (torch.randn(C, generator=g) * 0.2 + 1).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
        ss, cot = (torch.randn((N, 2 * C), generator=g) * 0.3).to(dev), torch.randn((N, H, W, C), generator=g).to(dev)
        runs = []
        for _ in range(3):
            leaves = [t.clone().requires_grad_(True) for t in (x, gamma, beta, ss)]
            (ut._GroupNormAct.apply(*leaves, True) * cot).sum().backward()
            runs.append([t.grad.clone() for t in leaves])
        for r in runs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(runs[0], r)), (N, C, H, W)


def test_training_step_gradients_are_bit_reproducible():
    """Every kernel of the training step sums in a fixed order: two backward passes over the same batch give identical parameter gradients."""
    from tests.test_train_loss_cpu import inputs, tiny_model
    model, diffusion = tiny_model()
    model = model.to(dev).train()
    x0, xc = (t.to(dev) for t in inputs())
    noise = torch.randn(x0.shape, generator=torch.Generator().manual_seed(5)).to(dev)
    grads = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        losses = diffusion.training_losses(model, x0.clamp(-1, 1), xc, torch.tensor([999, 17], device=dev),
                                           model_kwargs={"y": torch.tensor([3, 0], device=dev)}, noise=noise)
        losses["loss"].mean().backward()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    diff = [k for k in grads[0] if not torch.equal(grads[0][k], grads[1][k])]
    assert not diff, diff


@pytest.mark.parametrize("N,T,C,heads", [
    (2, 256, 128, 4),     # head size 32
    (2, 1024, 384, 4),    # production: 32x32 level, head size 96
    (2, 64, 768, 4),      # production: 8x8 level, head size 192
    (1, 256, 768, 4),     # production: 16x16 level
    (2, 40, 128, 4),      # T not a multiple of the 32-row tiles (masked rows / keys)
    (2, 16, 64, 4),       # head size 16 (the tiny test networks): the plain path through scratch
    (1, 64, 320, 2),      # head size 160: a multiple of 32 WITHOUT a register-resident kernel pair - scratch sized for the row kernels
    (2, 96, 256, 4),      # head size 64
    (1, 128, 256, 2),     # head size 128
])
def test_attention_backward_matches_torch_autograd(N, T, C, heads, monkeypatch):
    """hl_attention_nhwc_backward (csrc/hl_attention_bwd.hip) against float64 autograd of the reference's QKVAttention arithmetic
    (unet.py:255-274); no library GEMM is involved (torch.bmm / matmul raise while it runs) and two runs give the same bits."""
    from humanliff_amd.improved_diffusion import unet_train as ut
    g = torch.Generator().manual_seed(9 + T)
    qkv = torch.randn((N, 3 * C, T), generator=g)
    cot = torch.randn((N, C, T), generator=g)
    ch = C // heads
    qr = qkv.double().requires_grad_(True)
    q, k, v = qr.reshape(N * heads, 3 * ch, T).split(ch, dim=1)
    s = 1.0 / (ch ** 0.25)
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
    out = torch.einsum("bts,bcs->bct", wgt, v).reshape(N, C, T)
    (out * cot.double()).sum().backward()
    ref = qr.grad.permute(0, 2, 1)

    def boom(*a, **k):
        raise AssertionError("a library GEMM was called inside the HIP attention")
    grads = []
    for _ in range(2):
        qd = qkv.permute(0, 2, 1).contiguous().to(dev).requires_grad_(True)
        cd = cot.permute(0, 2, 1).contiguous().to(dev)
        with monkeypatch.context() as mp_:
            for name in ("bmm", "matmul", "baddbmm", "einsum"):
                mp_.setattr(torch, name, boom)
            o = ut._Attention.apply(qd, heads)
            (o * cd).sum().backward()
        grads.append(qd.grad.clone())
    assert (o.detach().cpu().permute(0, 2, 1).double() - out.detach()).abs().max() < 2e-5
    assert (grads[0].cpu().double() - ref).abs().max() < 2e-5 * float(ref.abs().max())
    assert torch.equal(grads[0], grads[1])                         # fixed summation order


def test_training_losses_backward_matches_reference_on_hip():
    """GaussianDiffusion.training_losses -> backward() through UNetModel.forward on the GPU: loss and parameter gradients equal the
    REFERENCE's (improved_diffusion imported unmodified by tests/golden/gen_golden_train_loss.py), and the PyTorch-op twin is never
    entered."""
    from tests.test_train_loss_cpu import inputs, tiny_model
    g = np.load(os.path.join(GOLDEN, "train_loss_tiny32.npz"))
    model, diffusion = tiny_model()
    model = model.to(dev).train()
    x0, xc = (t.to(dev) for t in inputs())

    # (the PyTorch-op twin lives under tests/: the product package cannot reach it - tests/test_train_loss_cpu.py::test_samplers_never_use_the_torch_twin)
    if True:
        losses = diffusion.training_losses(model, x0.clamp(-1, 1), xc, torch.tensor([999, 17], device=dev),
                                           model_kwargs={"y": torch.tensor([3, 0], device=dev)}, noise=torch.from_numpy(g["noise"]).to(dev))
        assert losses["loss"].requires_grad
        assert np.abs(losses["loss"].detach().cpu().numpy() - g["loss"]).max() < 1e-5
        losses["loss"].mean().backward()
    sd = dict(model.named_parameters())
    assert all(p.grad is not None for p in sd.values())
    tot = sum(float(p.grad.double().abs().sum()) for p in sd.values())
    assert abs(tot - float(g["grad_abs_sum"])) < 2e-4 * float(g["grad_abs_sum"])
    worst = 0.0
    for k in g["keys"]:
        ref = torch.from_numpy(g["g_" + str(k)])
        err = float((sd[str(k)].grad.cpu() - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        assert err < 2e-4, (str(k), err)
    print(f"training_losses on HIP: loss max-abs {np.abs(losses['loss'].detach().cpu().numpy() - g['loss']).max():.2e}, "
          f"worst relative gradient error over {len(g['keys'])} reference tensors {worst:.2e}, sum|grad| rel {abs(tot - float(g['grad_abs_sum'])) / float(g['grad_abs_sum']):.2e}")


@pytest.mark.parametrize("fused", [False, True])
def test_ddp_style_wrapper_trains_on_hip(fused):
    """train_util.py:236 hands training_losses the DDP-wrapped model: a wrapper whose forward calls the module must work and step.
    fused=True: torch's fused optimizers update parameters without bumping Tensor._version - the sampling path must still see the
    new weights (its packed copy is re-laid after a training forward)."""
    from tests.test_train_loss_cpu import inputs, tiny_model

    class Wrapper(torch.nn.Module):          # the call pattern of DistributedDataParallel: forward(*a, **k) -> self.module(*a, **k)
        def __init__(self, module):
            super().__init__()
            self.module = module
            self.calls = 0

        def forward(self, *a, **k):
            self.calls += 1
            return self.module(*a, **k)

    model, diffusion = tiny_model()
    model.to(dev)
    x0, xc = (t.to(dev) for t in inputs())
    t, y = torch.tensor([999, 17], device=dev), torch.tensor([3, 0], device=dev)
    model.eval()
    with torch.no_grad():                    # a sampling call BEFORE training: the packed copy of the initial weights exists
        out_before = model(x0, t, xc, y=y)
    wrapped = Wrapper(model.train())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=fused)
    noise = torch.randn(x0.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    vals = []
    for _ in range(5):
        loss = diffusion.training_losses(wrapped, x0.clamp(-1, 1), xc, t, model_kwargs={"y": y}, noise=noise)["loss"].mean()
        loss.backward()
        opt.step()
        opt.zero_grad()
        vals.append(float(loss.detach()))
    assert wrapped.calls == 5 and vals[-1] < vals[0]
    model.eval()
    with torch.no_grad():                    # the updated weights are picked up by the inference path
        out = model(x0, t, xc, y=y)
        twin = forward_autograd(model, x0, t, xc, y=y)        # PyTorch ops on the parameters as they are now
    assert torch.isfinite(out).all()
    assert (out - twin).abs().max() < 1e-4 * max(1.0, float(twin.abs().max())), float((out - twin).abs().max())
    assert (out - out_before).abs().max() > 1e-3             # ... and they did move


@pytest.mark.parametrize("cond,cin", [("concat", 54), ("AdaGN", 27)])
def test_training_other_cond_types_on_hip(cond, cin):
    """cond_type='concat' / 'AdaGN' through the product training path (training_losses -> model(...) -> forward_train): UNetModel.forward
    joins x and x_cond for 'concat' BEFORE handing over (unet.py:572-573) - forward_train must not join them again - and 'AdaGN' adds the
    projected condition to the embedding.  Loss and every parameter gradient against the PyTorch-op statement of the same network
    (forward_autograd, pinned to the reference's forward by tests/test_e2e_gpu.py::test_unet_cond_types_match_reference / _adagn_)."""
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults
    from humanliff_amd import synthetic as syn
    a = model_and_diffusion_defaults()
    size = 256 if cond == "AdaGN" else 32                 # AdaGN's Linear(64*64, E) fixes the input at 256x256 (narrow net there)
    a.update(dict(in_channels=cin, out_channels=27, class_cond=True, learn_sigma=False, num_heads=2, use_scale_shift_norm=True,
                  cond_type=cond, rescale_timesteps=False, dropout=0.0, image_size=size, num_channels=32, num_res_blocks=1,
                  attention_resolutions="16,8" if size == 32 else "32,16,8"))
    model, diffusion = create_model_and_diffusion(**a)
    ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict(syn.state_from_shapes(ks, 1), strict=True)
    model = model.to(dev).train()
    g = torch.Generator().manual_seed(21)
    B = 1 if size == 256 else 2
    x0 = torch.randn((B, 27, size, size), generator=g).clamp(-1, 1).to(dev)
    xc = (torch.randn((B, 27, size, size), generator=g).clamp(-1, 1) * 0.7).to(dev)
    noise = torch.randn((B, 27, size, size), generator=g).to(dev)
    t, y = torch.tensor([999, 17][:B], device=dev), torch.tensor([3, 0][:B], device=dev)
    loss = diffusion.training_losses(model, x0, xc, t, model_kwargs={"y": y}, noise=noise)["loss"]
    assert loss.requires_grad
    loss.mean().backward()
    got = {k: p.grad.clone() for k, p in model.named_parameters()}
    assert all(v is not None for v in got.values())
    model.zero_grad(set_to_none=True)
    twin = lambda x, ts, x_cond=None, y=None: forward_autograd(model, x, ts, x_cond, y)  # noqa: E731
    loss_t = diffusion.training_losses(twin, x0, xc, t, model_kwargs={"y": y}, noise=noise)["loss"]
    loss_t.mean().backward()
    assert float((loss.detach() - loss_t.detach()).abs().max()) < 1e-5 * max(1.0, float(loss_t.detach().abs().max()))
    worst = 0.0
    gscale = max(float(p.grad.abs().max()) for p in model.parameters())
    for k, p in model.named_parameters():
        ref = p.grad
        # relative to the tensor's own largest entry, with a floor: the bias of a convolution that feeds a GroupNorm has a gradient of
        # exactly zero in exact arithmetic (the mean subtraction removes it) - both sides hold rounding noise there
        scale = max(float(ref.abs().max()), 1e-4 * gscale)
        err = float((got[k] - ref).abs().max()) / scale
        worst = max(worst, err)
        # two fp32 evaluations with different summation orders (here: the twin runs on MIOpen / rocBLAS); measured worst 7e-4 on a bias
        # gradient (a sum over 2 x 32 x 32 pixels); the reference-pinned test above holds the HIP path to 2e-4 on the controlnet net
        assert err < 3e-3, (k, err)
    print(f"cond_type={cond}: HIP training path vs PyTorch-op twin, worst relative gradient error {worst:.2e}")


def test_sampling_between_forward_and_fused_step_sees_new_weights():
    """A preview sample drawn BETWEEN the training forward and the optimizer step re-packs the inference weights (and clears the stale
    mark); the backward pass that follows marks them stale again, so sampling after a fused optimizer step (which does not bump
    Tensor._version) still sees the new parameters."""
    from tests.test_train_loss_cpu import inputs, tiny_model
    model, diffusion = tiny_model()
    model.to(dev).train()
    x0, xc = (t.to(dev) for t in inputs())
    t, y = torch.tensor([999, 17], device=dev), torch.tensor([3, 0], device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, fused=True)
    loss = diffusion.training_losses(model, x0.clamp(-1, 1), xc, t, model_kwargs={"y": y})["loss"].mean()
    with torch.no_grad():
        preview = model(x0, t, xc, y=y)          # inference kernels; packs the CURRENT weights
    loss.backward()
    opt.step()
    with torch.no_grad():
        after = model(x0, t, xc, y=y)
        twin = forward_autograd(model, x0, t, xc, y=y)
    assert (after - preview).abs().max() > 1e-3               # the step moved the weights and sampling sees it
    assert (after - twin).abs().max() < 1e-4 * max(1.0, float(twin.abs().max()))


def test_eval_mode_input_gradient_is_not_dropped():
    """Guidance / cond_fn style callers ask for d out / d x on an eval() model: the differentiable HIP path answers (the inference
    kernels would return a tensor without grad_fn)."""
    from tests.test_train_loss_cpu import inputs, tiny_model
    model, _ = tiny_model()
    model.to(dev).eval()
    x0, xc = (t.to(dev) for t in inputs())
    t, y = torch.tensor([999, 17], device=dev), torch.tensor([3, 0], device=dev)
    x = x0.clone().requires_grad_(True)
    out = model(x, t, xc, y=y)
    assert out.requires_grad
    (gx,) = torch.autograd.grad(out.square().sum(), x)
    xt = x0.clone().requires_grad_(True)
    (gt,) = torch.autograd.grad(forward_autograd(model, xt, t, xc, y=y).square().sum(), xt)
    assert (gx - gt).abs().max() < 5e-4 * float(gt.abs().max())
    with torch.no_grad():                                     # and a plain sampling call on the same model still takes the inference kernels
        assert not model(x0, t, xc, y=y).requires_grad


@pytest.mark.parametrize("kind", ["bf16", "fp16"])
def test_16bit_conv_modes_forward_and_backward_data(kind):
    """HL_CONV_BF16 / HL_CONV_FP16 (16-bit MFMA arithmetic of the training path): operands rounded to 16 bits, fp32 accumulation.  3x3 layers
    (k_conv_h16: 16x16-pixel x 192-channel workgroups) and a 1x1 layer (k_conv1_h16 where it fills 48 workgroups, else k_conv_bf3 in the bf16 mode / the fp32 kernel in the fp16 mode), forward
    and backward-data, against float64: the relative error of one rounding per operand (2^-9 bf16, 2^-12 fp16), averaged down over K."""
    from humanliff_amd.improved_diffusion import unet_train as ut
    g = torch.Generator().manual_seed(4)
    lo, hi = (1e-5, 4e-3) if kind == "bf16" else (2e-6, 5e-4)
    for (N, H, W, C, Co, ks) in ((3, 64, 64, 96, 192, 3), (4, 32, 48, 64, 384, 3), (2, 64, 64, 384, 192, 1)):     # (3x3: 48 workgroups, where the dispatch starts to take k_conv_h16)
        x = torch.randn((N, C, H, W), generator=g)
        w = torch.randn((Co, C, ks, ks), generator=g) / (C * ks * ks) ** 0.5
        b = torch.randn((Co,), generator=g) * 0.1
        cot = torch.randn((N, Co, H, W), generator=g)
        xr = x.double().requires_grad_(True)
        yr = F.conv2d(xr, w.double(), b.double(), padding=ks // 2)
        (yr * cot.double()).sum().backward()
        xd = nhwc(x).to(dev).requires_grad_(True)
        wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        ut.set_train_arithmetic(kind)
        try:
            y = ut._Conv.apply(xd, wd, bd, 1, 0)
            (y * nhwc(cot).to(dev)).sum().backward()
        finally:
            ut.set_train_arithmetic(None)
        rel = lambda a, r: float((a.double() - r).norm() / r.norm())  # noqa: E731
        e_y, e_dx = rel(nchw(y.detach().cpu()), yr.detach()), rel(nchw(xd.grad.cpu()), xr.grad)
        ut.set_train_arithmetic("fp32")
        try:
            y32 = ut._Conv.apply(xd.detach(), wd.detach(), bd.detach(), 1, 0)
        finally:
            ut.set_train_arithmetic(None)
        e32 = rel(nchw(y32.cpu()), yr.detach())
        print(f"{C}->{Co} {ks}x{ks}: {kind} mode rel-L2 forward {e_y:.2e}, backward-data {e_dx:.2e} (fp32 mode forward {e32:.1e})")
        assert e32 < 1e-5
        hi_b = 4e-3                                                  # the backward of both modes runs in bf16 (gradients need fp32's exponent range)
        if ks == 1 and kind == "fp16":                               # 1x1 layers: k_conv1_h16 from 48 workgroups on (here the backward-data), else the fp32 kernel
            assert e_y < hi and 1e-6 < e_dx < hi_b
            continue
        assert lo < e_y < hi and e_dx < hi_b
        assert not torch.equal(y.detach(), y32)                      # the mode really switched arithmetic


@pytest.mark.parametrize("kind", ["bf16", "fp16"])
def test_16bit_weight_gradient_equals_the_gradient_of_the_rounded_operands(kind):
    """k_conv_wgrad_h16 (3x3 / stride-1 layers in the 16-bit training modes): dW is the float64 weight gradient of the input and the output
    gradient rounded to 16 bits, up to fp32 summation; db is the fp32 row sum of the unrounded output gradient.  Ragged tiles (sizes that
    are not multiples of 8), ragged channel blocks, several images, K slabs; bit-reproducible."""
    from humanliff_amd.improved_diffusion import unet_train as ut
    dt = torch.bfloat16                                               # (the backward of the fp16 mode runs in bf16 too: gradients need fp32's exponent range)
    for (N, H, W, C, Co, ups) in ((2, 32, 32, 64, 64, 0), (1, 20, 12, 96, 160, 0), (2, 64, 64, 192, 192, 0), (3, 8, 8, 32, 32, 0), (2, 16, 24, 64, 96, 1)):
        g = torch.Generator().manual_seed(H + C)
        x = torch.randn((N, C, H, W), generator=g)
        w = torch.randn((Co, C, 3, 3), generator=g) / (C * 9) ** 0.5
        b = torch.randn((Co,), generator=g)
        cot = torch.randn((N, Co, H * (2 if ups else 1), W * (2 if ups else 1)), generator=g)
        wr = w.double().requires_grad_(True)
        xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
        (F.conv2d(xin.to(dt).double(), wr, None, padding=1) * cot.to(dt).double()).sum().backward()
        dw_ref, db_ref = wr.grad, cot.double().sum(dim=(0, 2, 3))
        runs = []
        for _ in range(2):
            xd = nhwc(x).to(dev).requires_grad_(True)
            wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
            ut.set_train_arithmetic(kind)
            try:
                (ut._Conv.apply(xd, wd, bd, 1, ups) * nhwc(cot).to(dev)).sum().backward()
            finally:
                ut.set_train_arithmetic(None)
            runs.append((wd.grad.clone(), bd.grad.clone()))
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
        dw, db = runs[0]
        ew = float((dw.cpu().double() - dw_ref).abs().max() / dw_ref.abs().max())
        eb = float((db.cpu().double() - db_ref).abs().max() / db_ref.abs().max())
        print(f"{kind} N{N} {H}x{W} {C}->{Co} ups{ups}: dW max-abs / max {ew:.2e}, db {eb:.2e}")
        assert ew < 2e-5 and eb < 1e-5, (N, H, W, C, Co, ew, eb)


@pytest.fixture
def h16_single_tiles():
    """The dispatch takes k_conv_h16 from 48 workgroups on; these unit tests run it on single tiles too (hl_debug_set_h16_min_blocks)."""
    from humanliff_amd import _lib
    L = _lib.lib()
    L.hl_debug_set_h16_min_blocks(1)
    yield
    L.hl_debug_set_h16_min_blocks(-1)


@pytest.mark.parametrize("f16", [0, 1])
def test_conv_h16_equals_the_convolution_of_the_rounded_operands(f16, h16_single_tiles):
    """k_conv_h16 through the C ABI (hl_conv2d_nhwc_mode / hl_conv2d_nhwc_gn): the kernel's result is the float64 convolution of the operands
    rounded to 16 bits, up to fp32 summation - ragged tile counts, a residual, the GroupNorm + SiLU pre-pass, the emitted GroupNorm statistics."""
    from humanliff_amd import _lib
    L = _lib.lib()
    mode, dt = (_lib.HL_CONV_FP16, torch.float16) if f16 else (_lib.HL_CONV_BF16, torch.bfloat16)
    for (N, H, W, C, Co, use_res, gn, ups) in ((1, 16, 16, 32, 192, 0, 0, 0), (2, 48, 80, 64, 384, 1, 0, 0), (3, 32, 16, 96, 192, 1, 1, 0), (2, 16, 24, 64, 192, 1, 0, 1)):
        g = torch.Generator().manual_seed(N + C)
        x = torch.randn((N, H, W, C), generator=g); w = torch.randn((Co, C, 3, 3), generator=g) / (C * 9) ** 0.5; b = torch.randn(Co, generator=g)
        Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
        res = torch.randn((N, Ho, Wo, Co), generator=g)
        cA, cB = torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g) * 0.1
        xin = x
        if gn:
            u = x * cA[:, None, None, :] + cB[:, None, None, :]
            xin = (u * torch.sigmoid(u)).to(dev).cpu()               # (the pre-pass runs in fp32 on the GPU; its rounding is inside the tolerance)
        xr = xin.to(dt).double().permute(0, 3, 1, 2)
        if ups:
            xr = F.interpolate(xr, scale_factor=2, mode="nearest")
        ref = F.conv2d(xr, w.to(dt).double(), b.double(), padding=1).permute(0, 2, 3, 1)
        if use_res:
            ref = ref + res.double()
        xd, wd, bd, rd, ad, bd2 = (t.to(dev) for t in (x, w, b, res, cA, cB))
        out = torch.zeros((N, Ho, Wo, Co), device=dev)
        scratch = torch.empty(Co * C * 9 * 6 + 256 + (8 << 20) + N * H * W * C, device=dev)
        with _lib.on(dev):
            _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(xd), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Co, 3, 1, ups, _lib.ptr(ad) if gn else None,
                                             _lib.ptr(bd2) if gn else None, gn, _lib.ptr(rd) if use_res else None, _lib.ptr(out), _lib.ptr(scratch),
                                             scratch.numel() * 4, _lib.stream_ptr()), "hl_conv2d_nhwc_mode")
        err = float((out.cpu().double() - ref).abs().max())
        assert err < (2e-5 if not gn else 2e-3), (N, H, W, C, Co, err)       # (with the pre-pass: a value next to a rounding boundary may round the other way)
        assert float((out.cpu().double() - ref).norm() / ref.norm()) < (1e-6 if not gn else 1e-4)


def test_training_under_autocast_and_in_bf16_arithmetic_tracks_fp32():
    """The reference trains under torch.autocast (train_util.py:214, --use_amp True).  The HIP training path accepts the autocast context
    (its kernels take fp32 tensors, so torch's own autocasting is switched off inside) and computes its convolutions with operands of the
    autocast dtype (k_conv_h16 where it applies), exactly as `set_train_arithmetic("bf16")` does by hand; `set_train_arithmetic("fp32")`
    pins the exact kernels.  Loss, gradients and a 20-step loss curve of the 16-bit runs track the fp32 run."""
    from tests.test_train_loss_cpu import inputs, tiny_model
    from humanliff_amd.improved_diffusion import unet_train as ut
    x0, xc = (t.to(dev) for t in inputs())
    t, y = torch.tensor([999, 17], device=dev), torch.tensor([3, 0], device=dev)
    noise = torch.randn(x0.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))

    def wide_model():       # 192 base channels: layers wide enough for the direct DMA tile, which is where HL_CONV_BF16 applies
        from humanliff_amd import synthetic as syn
        from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults
        a = model_and_diffusion_defaults()
        a.update(dict(in_channels=27, out_channels=27, class_cond=True, learn_sigma=False, num_heads=4, use_scale_shift_norm=True,
                      cond_type="controlnet", rescale_timesteps=False, dropout=0.0, diffusion_steps=1000, noise_schedule="linear",
                      timestep_respacing="", image_size=32, num_channels=192, num_res_blocks=1, attention_resolutions="16"))
        model, diffusion = create_model_and_diffusion(**a)
        ks = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        model.load_state_dict(syn.state_from_shapes(ks, 1), strict=True)
        return model, diffusion

    def run(amp, arith, steps):
        model, diffusion = wide_model()
        model.to(dev).train()
        opt = torch.optim.SGD(model.parameters(), lr=0.02)    # (plain SGD: Adam's per-parameter normalisation turns rounding noise on tiny gradients into full-size steps)
        losses, first_grads = [], None
        ut.set_train_arithmetic(arith)
        try:
            for i in range(steps):
                with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=amp):
                    loss = diffusion.training_losses(model, x0.clamp(-1, 1), xc, t, model_kwargs={"y": y}, noise=noise)["loss"].mean()
                assert loss.dtype == torch.float32
                loss.backward()
                if i == 0:
                    first_grads = {k: p.grad.clone() for k, p in model.named_parameters()}
                opt.step()
                opt.zero_grad()
                losses.append(float(loss.detach()))
        finally:
            ut.set_train_arithmetic(None)
        return losses, first_grads

    l32, g32 = run(False, None, 20)
    lam, gam = run(True, None, 20)            # under autocast(bfloat16): the convolutions take the autocast dtype = the "bf16" arithmetic
    l16, g16 = run(False, "bf16", 20)         # the same arithmetic chosen by hand
    assert lam == l16 and all(torch.equal(gam[k], g16[k]) for k in g16)     # (every kernel of the step sums in a fixed order: identical runs)
    lpin, _ = run(True, "fp32", 3)            # a pinned arithmetic wins over the autocast context
    assert lpin == l32[:3]
    assert l32[0] != l16[0]                                          # different arithmetic ...
    assert abs(l16[0] - l32[0]) < 2e-3 * abs(l32[0])                 # ... same loss to bf16 accuracy
    num = sum(float(((g16[k] - g32[k]).double() ** 2).sum()) for k in g32)
    den = sum(float((g32[k].double() ** 2).sum()) for k in g32)
    print(f"bf16 arithmetic: first loss {l16[0]:.6f} vs fp32 {l32[0]:.6f}; gradient rel-L2 over all parameters {(num / den) ** 0.5:.2e}; "
          f"loss after 20 steps {l16[-1]:.5f} vs {l32[-1]:.5f}")
    assert (num / den) ** 0.5 < 3e-2
    assert l32[-1] < l32[0] and l16[-1] < l16[0]
    print("loss curves fp32 / bf16:", [round(v, 4) for v in l32[::4]], [round(v, 4) for v in l16[::4]])
    worst = max(abs(a - b) / abs(b) for a, b in zip(l16, l32))
    assert worst < 0.05, worst


def test_no_scale_shift_norm_and_dropout_on_hip():
    """use_scale_shift_norm=False (the embedding is added before out_layers' GroupNorm, unet.py:216-218) on the HIP inference AND training
    paths against the reference's forward, loss and gradients (tests/golden/gen_golden_noss.py); dropout > 0 (unet.py:196) acts in the
    training path only."""
    from tests.test_train_loss_cpu import inputs, tiny_model
    g = np.load(os.path.join(GOLDEN, "unet_noss.npz"))
    model, diffusion = tiny_model(use_scale_shift_norm=False)
    model = model.to(dev).eval()
    x0, xc = (t.to(dev) for t in inputs())
    t, y = torch.tensor([999, 17], device=dev), torch.tensor([3, 0], device=dev)
    with torch.no_grad():
        out = model(x0, t, xc, y=y).cpu()
    assert (out - torch.from_numpy(g["out"])).abs().max() < 2e-5
    model.train()

    # (the PyTorch-op twin lives under tests/: the product package cannot reach it - tests/test_train_loss_cpu.py::test_samplers_never_use_the_torch_twin)
    if True:
        losses = diffusion.training_losses(model, x0.clamp(-1, 1), xc, t, model_kwargs={"y": y}, noise=torch.from_numpy(g["noise"]).to(dev))
        assert np.abs(losses["loss"].detach().cpu().numpy() - g["loss"]).max() < 1e-5
        losses["loss"].mean().backward()
    sd = dict(model.named_parameters())
    tot = sum(float(p.grad.double().abs().sum()) for p in sd.values())
    assert abs(tot - float(g["grad_abs_sum"])) < 2e-4 * float(g["grad_abs_sum"])
    for k in g["keys"]:
        ref = torch.from_numpy(g["g_" + str(k)])
        assert float((sd[str(k)].grad.cpu() - ref).abs().max()) < 2e-4 * float(ref.abs().max()) + 1e-8, str(k)     # (floor: some gradients of this net are ~1e-10)
    # dropout: eval-mode sampling ignores it; the training path draws a new mask per call
    md, _ = tiny_model(dropout=0.5)
    md = md.to(dev)
    m0, _ = tiny_model(dropout=0.0)
    m0 = m0.to(dev).eval()
    with torch.no_grad():
        assert torch.equal(md.eval()(x0, t, xc, y=y), m0(x0, t, xc, y=y))
    md.train()
    torch.manual_seed(0)
    a = md(x0.requires_grad_(False), t, xc, y=y)
    b = md(x0, t, xc, y=y)
    assert a.requires_grad and not torch.equal(a, b)


@pytest.mark.parametrize("f16", [0, 1])
def test_conv1_h16_equals_the_product_of_the_rounded_operands(f16, h16_single_tiles):
    """k_conv1_h16 (1x1 layers of the 16-bit modes: 256 pixels x 192 channels per workgroup, chunks of 96 input channels) through the C ABI:
    the float64 product of the operands rounded to 16 bits up to fp32 summation; residual, GroupNorm pre-pass (no SiLU: the qkv convolution),
    a channel pitch on the input (a slice of a concat buffer is what the decoder's skip convolutions read)."""
    from humanliff_amd import _lib
    L = _lib.lib()
    mode, dt = (_lib.HL_CONV_FP16, torch.float16) if f16 else (_lib.HL_CONV_BF16, torch.bfloat16)
    for (N, H, W, C, Co, use_res, gn) in ((2, 16, 16, 96, 192, 0, 0), (1, 32, 32, 384, 384, 1, 0), (2, 16, 32, 192, 576, 1, 1), (1, 16, 16, 1344, 192, 0, 0)):
        g = torch.Generator().manual_seed(N + C)
        x = torch.randn((N, H, W, C), generator=g); w = torch.randn((Co, C, 1, 1), generator=g) / C ** 0.5; b = torch.randn(Co, generator=g)
        res = torch.randn((N, H, W, Co), generator=g)
        cA, cB = torch.rand((N, C), generator=g) + 0.5, torch.randn((N, C), generator=g) * 0.1
        xin = (x * cA[:, None, None, :] + cB[:, None, None, :]).to(dev).cpu() if gn else x
        ref = torch.einsum("nhwc,oc->nhwo", xin.to(dt).double(), w[:, :, 0, 0].to(dt).double()) + b.double()
        if use_res:
            ref = ref + res.double()
        xd, wd, bd, rd, ad, bd2 = (t.to(dev) for t in (x, w, b, res, cA, cB))
        out = torch.zeros((N, H, W, Co), device=dev)
        scratch = torch.empty(Co * C * 6 + 256 + (8 << 20) + N * H * W * C, device=dev)
        with _lib.on(dev):
            _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(xd), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Co, 1, 1, 0, _lib.ptr(ad) if gn else None,
                                             _lib.ptr(bd2) if gn else None, 0, _lib.ptr(rd) if use_res else None, _lib.ptr(out), _lib.ptr(scratch),
                                             scratch.numel() * 4, _lib.stream_ptr()), "hl_conv2d_nhwc_mode")
        err = float((out.cpu().double() - ref).abs().max())
        assert err < (3e-5 if not gn else 3e-3), (N, H, W, C, Co, err)
        assert float((out.cpu().double() - ref).norm() / ref.norm()) < (1e-6 if not gn else 1e-4)


def test_fp16_mode_keeps_tiny_gradients():
    """Gradients of 1e-8 are below fp16's subnormal range; the backward of the fp16 mode runs in bf16 (fp32's exponents), so they survive
    without loss scaling: backward-data and weight gradient of a 3x3 layer stay at bf16 accuracy for an output gradient scaled by 1e-8."""
    from humanliff_amd.improved_diffusion import unet_train as ut
    g = torch.Generator().manual_seed(12)
    N, C, Co, H, W = 3, 192, 192, 64, 64
    x = torch.randn((N, C, H, W), generator=g)
    w = torch.randn((Co, C, 3, 3), generator=g) / (C * 9) ** 0.5
    cot = torch.randn((N, Co, H, W), generator=g) * 1e-8
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    (F.conv2d(xr, wr, None, padding=1) * cot.double()).sum().backward()
    xd, wd = nhwc(x).to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    ut.set_train_arithmetic("fp16")
    try:
        (ut._Conv.apply(xd, wd, None, 1, 0) * nhwc(cot).to(dev)).sum().backward()
    finally:
        ut.set_train_arithmetic(None)
    rel = lambda a, r: float((a.double() - r).norm() / r.norm())  # noqa: E731
    e_dx, e_dw = rel(nchw(xd.grad.cpu()), xr.grad), rel(wd.grad.cpu(), wr.grad)
    print(f"output gradient ~1e-8: backward-data rel-L2 {e_dx:.2e}, weight gradient rel-L2 {e_dw:.2e}")
    assert e_dx < 5e-3 and e_dw < 5e-3


@pytest.mark.parametrize("f16", [0, 1])
def test_conv_h16_split_k_on_small_layers(f16):
    """3x3 layers with fewer than 48 tiles x channel blocks split their input channels into slabs (k_conv_h16 with blockIdx.z, raw sums to the
    split-K workspace, k_splitk_finish adds them in a fixed order with bias / residual): still the convolution of the rounded operands."""
    from humanliff_amd import _lib
    L = _lib.lib()
    mode, dt = (_lib.HL_CONV_FP16, torch.float16) if f16 else (_lib.HL_CONV_BF16, torch.bfloat16)
    for (N, H, W, C, Co, use_res) in ((2, 32, 32, 576, 576, 1), (4, 16, 16, 768, 768, 0), (1, 32, 32, 1152, 192, 1)):
        g = torch.Generator().manual_seed(N + C)
        x = torch.randn((N, H, W, C), generator=g); w = torch.randn((Co, C, 3, 3), generator=g) / (C * 9) ** 0.5; b = torch.randn(Co, generator=g)
        res = torch.randn((N, H, W, Co), generator=g)
        ref = F.conv2d(x.to(dt).double().permute(0, 3, 1, 2), w.to(dt).double(), b.double(), padding=1).permute(0, 2, 3, 1)
        if use_res:
            ref = ref + res.double()
        xd, wd, bd, rd = (t.to(dev) for t in (x, w, b, res))
        outs = []
        for _ in range(2):
            out = torch.zeros((N, H, W, Co), device=dev)
            scratch = torch.empty(Co * C * 9 * 6 + 256 + (16 << 20), device=dev)
            with _lib.on(dev):
                _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(xd), N, H, W, C, _lib.ptr(wd), _lib.ptr(bd), Co, 3, 1, 0, None, None, 0,
                                                 _lib.ptr(rd) if use_res else None, _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()),
                           "hl_conv2d_nhwc_mode")
            outs.append(out)
        assert torch.equal(outs[0], outs[1])
        err = float((outs[0].cpu().double() - ref).abs().max())
        rel = float((outs[0].cpu().double() - ref).norm() / ref.norm())
        assert err < 5e-5 and rel < 2e-6, (N, H, W, C, Co, err, rel)      # (the fp32 kernels would sit at the operand rounding: rel 2e-4 / 2e-3)


def test_graphed_train_step_equals_eager_steps():
    """GraphedTrainStep (the whole step as one HIP graph) against the same steps issued from Python: same batches, timesteps and noise, a
    capturable fused AdamW on both sides - the kernels sum in a fixed order, so parameters and losses agree bit for bit; construction leaves
    parameters and optimizer state untouched, and a sampling call afterwards sees the updated weights."""
    from tests.test_train_loss_cpu import inputs, tiny_model
    from humanliff_amd.improved_diffusion.unet_train import GraphedTrainStep
    x0, xc = (t.to(dev) for t in inputs())
    x0 = x0.clamp(-1, 1)
    gen = torch.Generator().manual_seed(11)
    batches = [((x0 + 0.1 * torch.randn(x0.shape, generator=gen).to(dev)).clamp(-1, 1), torch.randint(0, 1000, (2,), generator=gen).to(dev),
                torch.randn(x0.shape, generator=gen).to(dev), torch.randint(0, 4, (2,), generator=gen).to(dev)) for _ in range(4)]

    def fresh():
        torch.manual_seed(0)
        model, diffusion = tiny_model()
        model = model.to(dev).train()
        return model, diffusion, torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.0, fused=True, capturable=True)

    m1, d1, o1 = fresh()
    eager = []
    for x, t, n, y in batches:
        o1.zero_grad(set_to_none=True)
        loss = d1.training_losses(m1, x, xc, t, model_kwargs={"y": y}, noise=n)["loss"].mean()
        loss.backward()
        o1.step()
        eager.append(float(loss.detach()))
    m2, d2, o2 = fresh()
    before = [p.detach().clone() for p in m2.parameters()]
    step = GraphedTrainStep(d2, m2, o2, batches[0][0], xc, batches[0][1], {"y": batches[0][3]}, with_noise=True)
    assert all(torch.equal(a, b) for a, b in zip(before, m2.parameters())), "construction must leave the parameters as they were"
    graphed = [float(step(x, xc, t, {"y": y}, noise=n)) for x, t, n, y in batches]
    assert graphed == eager, (graphed, eager)
    bad = [k for (k, a), b in zip(m1.named_parameters(), m2.parameters()) if not torch.equal(a, b)]
    assert not bad, bad
    m1.eval(), m2.eval()
    with torch.no_grad():
        t, y = batches[0][1], batches[0][3]
        assert torch.equal(m1(x0, t, xc, y=y), m2(x0, t, xc, y=y))       # the packed inference weights were re-laid after the replayed steps


def test_graphed_train_step_takes_over_from_eager_steps():
    """Capture in the middle of a run: one eager step (its loss dropped - a live loss pins the AccumulateGrad nodes to its stream), then the
    graph for the remaining batches; the optimizer's running state must survive construction (warm-up iterations are rolled back) and the
    result equal the all-eager run."""
    from tests.test_train_loss_cpu import inputs, tiny_model
    from humanliff_amd.improved_diffusion.unet_train import GraphedTrainStep
    x0, xc = (t.to(dev) for t in inputs())
    gen = torch.Generator().manual_seed(12)
    batches = [((x0 + 0.1 * torch.randn(x0.shape, generator=gen).to(dev)).clamp(-1, 1), torch.randint(0, 1000, (2,), generator=gen).to(dev),
                torch.randn(x0.shape, generator=gen).to(dev), torch.randint(0, 4, (2,), generator=gen).to(dev)) for _ in range(3)]

    def fresh():
        torch.manual_seed(0)
        model, diffusion = tiny_model()
        model = model.to(dev).train()
        return model, diffusion, torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.0, fused=True, capturable=True)

    def eager(m, d, o, b):
        x, t, n, y = b
        o.zero_grad(set_to_none=True)
        d.training_losses(m, x, xc, t, model_kwargs={"y": y}, noise=n)["loss"].mean().backward()
        o.step()

    m1, d1, o1 = fresh()
    for b in batches:
        eager(m1, d1, o1, b)
    m2, d2, o2 = fresh()
    eager(m2, d2, o2, batches[0])
    step = GraphedTrainStep(d2, m2, o2, batches[1][0], xc, batches[1][1], {"y": batches[1][3]}, with_noise=True)
    for x, t, n, y in batches[1:]:
        step(x, xc, t, {"y": y}, noise=n)
    bad = [k for (k, a), b in zip(m1.named_parameters(), m2.parameters()) if not torch.equal(a, b)]
    assert not bad, bad
