"""UNetModel.forward in plain PyTorch ops, differentiable - the CPU-checkable statement of the network.

TEST INFRASTRUCTURE (it lives under tests/, outside the product package): sampling runs the fused HIP forward (unet.py ->
hl_unet_forward), training the HIP forward + backward behind autograd.Functions (unet_train.py).  It exists for the gradient tests: tests/test_train_loss_cpu.py pins it to the
reference's loss and parameter gradients on the CPU, tests/test_unet_train_gpu.py pins the HIP training path to the same vectors (with
this module patched to raise), and tests/test_unet_gpu.py checks that all three statements agree on the GPU; scripts/unet_train_bench.py
can time it (MIOpen / rocBLAS) next to the HIP path.

Same arithmetic as human_diffusion/improved_diffusion/unet.py:550-615 on the parameter-holder modules of unet.py: GroupNorm(32,
eps 1e-5) -> SiLU -> conv ResBlocks with scale-shift conditioning (:203-206), per-head [q|k|v] attention with ch^-1/4 on q and k and
fp32 softmax (:248-274), stride-2 conv downsampling, nearest x2 + conv upsampling, and the control encoder on x + x_cond whose
feature map is REPLACED by its zero-conv projection before the next block (:594-606).
"""
import math

import torch as th
import torch.nn.functional as F

from humanliff_amd.improved_diffusion import unet as U


def _embedding(timesteps, dim, max_period=10000):
    half = dim // 2
    freqs = th.exp(-math.log(max_period) * th.arange(half, dtype=th.float32, device=timesteps.device) / half)
    args = timesteps[:, None].float() * freqs[None]
    return th.cat([th.cos(args), th.sin(args)], dim=-1)


def _silu(x):
    return x * th.sigmoid(x)


def _norm(m, x):
    return F.group_norm(x.float(), m.num_groups, m.weight, m.bias, m.eps)


def _res_block(m, x, emb):
    h = m.in_layers[2](_silu(_norm(m.in_layers[0], x)))
    e = m.emb_layers[1](_silu(emb))
    if getattr(m, "use_scale_shift_norm", True):
        scale, shift = e.chunk(2, dim=1)
        h = _norm(m.out_layers[0], h) * (1 + scale[:, :, None, None]) + shift[:, :, None, None]
    else:
        h = _norm(m.out_layers[0], h + e[:, :, None, None])
    if getattr(m, "use_3d_aware", False):    # unet.py:208-214: each plane + the other two averaged along the axis it does not share
        w3 = h.shape[-1] // 3
        p0, p1, p2 = h[..., :w3], h[..., w3:2 * w3], h[..., 2 * w3:]
        row = lambda p: p.mean(-1, keepdim=True).expand(-1, -1, -1, w3)          # noqa: E731
        col = lambda p: p.mean(-2, keepdim=True).expand(-1, -1, p.shape[-2], -1)  # noqa: E731
        h = th.cat([th.cat([p0, row(p1), col(p2)], 1), th.cat([p1, row(p0), row(p2)], 1), th.cat([p2, col(p0), col(p1)], 1)], -1)
    h = m.out_layers[3](m.out_layers[2](_silu(h)))                             # (out_layers[2]: nn.Dropout)
    return m.skip_connection(x) + h


def _attention(m, x):
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = m.qkv(_norm(m.norm, xf))
    qkv = qkv.reshape(b * m.num_heads, -1, qkv.shape[2])
    ch = qkv.shape[1] // 3
    q, k, v = qkv.split(ch, dim=1)
    s = 1.0 / math.sqrt(math.sqrt(ch))
    w = th.softmax(th.einsum("bct,bcs->bts", q * s, k * s).float(), dim=-1)
    a = th.einsum("bts,bcs->bct", w, v).reshape(b, -1, hh * ww)
    return (xf + m.proj_out(a)).reshape(b, c, hh, ww)


def _xattn(a, x, context):
    """CrossAttention.forward (spatial_transformer.py:86-112), no mask."""
    ctx = x if context is None else context
    q, k, v = a.to_q(x), a.to_k(ctx), a.to_v(ctx)
    b, nq, _ = q.shape
    sp = lambda t: t.reshape(b, t.shape[1], a.heads, -1).permute(0, 2, 1, 3).reshape(b * a.heads, t.shape[1], -1)  # noqa: E731
    q, k, v = sp(q), sp(k), sp(v)
    w = th.softmax(th.einsum("bid,bjd->bij", q, k) * a.scale, dim=-1)
    o = th.einsum("bij,bjd->bid", w, v).reshape(b, a.heads, nq, -1).permute(0, 2, 1, 3).reshape(b, nq, -1)
    return a.to_out[0](o)


def _spatial_transformer(m, x, context):
    """SpatialTransformer.forward (spatial_transformer.py:165-178) with its BasicTransformerBlock (:128-134)."""
    b, c, hh, ww = x.shape
    h = m.proj_in(F.group_norm(x.float(), 32, m.norm.weight, m.norm.bias, m.norm.eps))
    h = h.reshape(b, c, hh * ww).permute(0, 2, 1)
    for blk in m.transformer_blocks:
        h = _xattn(blk.attn1, blk.norm1(h), None) + h
        h = _xattn(blk.attn2, blk.norm2(h), context) + h
        u, gate = blk.ff.net[0].proj(blk.norm3(h)).chunk(2, dim=-1)
        h = blk.ff.net[2](u * F.gelu(gate)) + h
    h = h.permute(0, 2, 1).reshape(b, c, hh, ww)
    return m.proj_out(h) + x


def _run(seq, h, emb, context=None):
    for m in seq:
        if isinstance(m, U.ResBlock):
            h = _res_block(m, h, emb)
        elif isinstance(m, U.SpatialTransformer):
            h = _spatial_transformer(m, h, context)
        elif isinstance(m, U.AttentionBlock):
            h = _attention(m, h)
        elif isinstance(m, U.Downsample):
            h = m.op(h)
        elif isinstance(m, U.Upsample):
            h = m.conv(F.interpolate(h, scale_factor=2, mode="nearest"))
        else:                                  # the bare first convolution of an encoder
            h = m(h)
    return h


def forward_autograd(model, x, timesteps, x_cond=None, y=None):
    """UNetModel.forward's contract (x (N,C,H,W), timesteps (N,), x_cond, y) with autograd; fp32."""
    if model.num_classes is not None:
        assert y is not None and y.shape == (x.shape[0],)
    aware = getattr(model, "use_3d_aware", False)
    if aware:                                # unet.py:566-570: the planes side by side - x and x_cond each, BEFORE a 'concat' joins them
        roll = lambda v: th.cat(v.chunk(3, dim=1), -1)  # noqa: E731
        x = roll(x)
        if x_cond is not None:
            x_cond = roll(x_cond)
    if model.cond_type == "concat" and x_cond is not None:   # unet.py:572-573
        x, x_cond = th.cat([x, x_cond], dim=1), None
    emb = model.time_embed[2](_silu(model.time_embed[0](_embedding(timesteps, model.model_channels))))
    context = None
    if model.cond_type in ("AdaGN", "cross_attention"):   # unet.py:574-582
        assert x_cond is not None, f"cond_type='{model.cond_type}' needs x_cond"
        xp = model.conv_proj_2(model.conv_proj_1(x_cond.float()))
        xp = model.linear(xp.reshape(xp.shape[0], -1))
        if model.cond_type == "AdaGN":
            emb = emb + xp
        else:
            context = xp.unsqueeze(1)
    if model.num_classes is not None:
        emb = emb + model.label_emb(y)
    hs = []
    h = x.float()
    for blk in model.input_blocks:
        h = _run(blk, h, emb, context)
        hs.append(h)
    h = _run(model.middle_block, h, emb, context)
    if model.cond_type == "controlnet":
        assert x_cond is not None, "cond_type='controlnet' needs x_cond (zeros for the first layer)"
        hs_cond = []
        hc = x.float() + x_cond.float()
        for blk, proj in zip(model.input_blocks_cond, model.input_blocks_proj_cond):
            hc = proj(_run(blk, hc, emb))
            hs_cond.append(hc)
    for blk in model.output_blocks:
        skip = hs.pop()
        if model.cond_type == "controlnet":
            skip = skip + hs_cond.pop()
        h = _run(blk, th.cat([h, skip], dim=1), emb, context)
    h = model.out[2](_silu(_norm(model.out[0], h)))
    if aware:                                # unet.py:613-614
        w3 = h.shape[-1] // 3
        h = th.cat([h[..., :w3], h[..., w3:2 * w3], h[..., 2 * w3:]], 1)
    return h.to(x.dtype)
