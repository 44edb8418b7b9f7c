"""Differentiable UNetModel.forward on the HIP kernels - the training path (UNet half).

What the reference trains with: GaussianDiffusion.training_losses (gaussian_diffusion.py:688-772) -> mse -> loss.backward() through
UNetModel.forward (unet.py:550-615), called by TrainLoop.forward_backward through the DDP wrapper (train_util.py:200-285).  Here the
same function is a chain of torch.autograd.Functions whose forward AND backward are the library's kernels (C ABI of
include/humanliff_hip.h), activations NHWC fp32:

    _Conv          forward   hl_conv2d_nhwc_mode (Winograd / direct fp32 MFMA kernels of the inference path)
                   d input   hl_conv2d_nhwc_bwd_data: the same forward kernels on the output gradient, the weights read flipped and
                             channel-transposed while they are re-laid (stride 2: zero-stuffed gradient first; nearest-x2 upsample:
                             2x2 block sums afterwards)
                   d weight  hl_conv2d_wgrad_nhwc_ws (pixels-as-K MFMA GEMM: 3x3 layers with all nine taps per workgroup from an LDS-staged
                             tile, 1x1 layers with 192 x 64 channel blocks; deterministic slab sums), bias gradient in the same launch
    _GroupNormAct  forward   hl_groupnorm_train_forward (statistics -> affine with scale/shift -> apply + SiLU)
                   backward  hl_groupnorm_train_backward (per-(n,c) reductions, the (N,C) algebra, dx; parameter / scale-shift gradients)
    _Attention     forward   hl_attention_nhwc (fp32 flash-style kernel)
                   backward  hl_attention_nhwc_backward (csrc/hl_attention_bwd.hip: fp32 MFMA, probabilities recomputed, deterministic; round 4 -
                             rounds 2-3 used five torch.bmm here)
The (N, 768)-sized embedding MLP (time_embed, label_emb, the ResBlocks' emb_layers: 0.004 % of the FLOPs), residual adds and channel
concatenations are torch tensor ops.  No convolution, normalisation or attention runs through MIOpen / torch.nn.functional.

UNetModel.forward takes this path when gradients are enabled on a model in training mode (unet.py); the samplers (no_grad, eval) never
do.  The PyTorch-op twin (tests/unet_autograd_twin.py, test infrastructure) remains as the CPU-checkable statement of the same function that the gradient tests compare
both against the reference's vectors.
"""
import ctypes as C
import math

import threading
import weakref

import torch as th
import torch.nn.functional as F

from .. import _lib
from . import unet as U

_MODE = _lib.HL_CONV_FP32
_ARITH = {"mode": None}                     # set_train_arithmetic's choice (process-wide, set by the user)
_TLS = threading.local()                    # .autocast: the dtype of the CALLING THREAD's autocast region (forward_train); DataParallel runs its
                                            # replicas in threads that inherit the autocast state - a module global would leak between them
_LOGGED = set()
_KIND_MODE = {"fp32": _lib.HL_CONV_FP32, "bf16": _lib.HL_CONV_BF16, "fp16": _lib.HL_CONV_FP16}


def set_train_arithmetic(kind=None):
    """Arithmetic of the convolutions of the training path (forward and backward-data).
      "fp32"          the exact fp32 kernels (Winograd where it applies);
      "fp16" / "bf16" 16-bit operands, fp32 accumulation (HL_CONV_FP16 / HL_CONV_BF16; the BACKWARD of "fp16" runs in bf16: gradients need
                      fp32's exponent range unless the loss is scaled): k_conv_h16 on the 3x3 / stride-1 layers it covers
                      (2.2-2.6x the fp32 Winograd kernel on the 256x256 layers), the other layers on k_conv_bf3 (bf16) or the fp32 kernels
                      (fp16); fp32 tensors and master weights;
      None (default)  "fp32", unless the caller trains under torch.autocast (train_util.py:214, --use_amp True): then the autocast dtype -
                      what the reference's AMP computes its convolutions in.
    The weight gradients of the 3x3 / stride-1 layers take the same arithmetic (k_conv_wgrad_h16); the other weight gradients, GroupNorm
    and attention stay fp32."""
    assert kind in (None, "fp32", "bf16", "fp16")
    _ARITH["mode"] = kind


def _conv_mode():
    kind = _ARITH["mode"]
    if kind is None:
        kind = getattr(_TLS, "autocast", None) or "fp32"
    return _MODE if kind == "fp32" else _KIND_MODE[kind]


def _conv_raw(x, w, b, ks, stride=1, ups=0, mode=None):
    """x (N,H,W,Cx) NHWC dense, Cx % 16 == 0; w (Cout, Cx, ks, ks) contiguous; -> (N,Ho,Wo,Cout)."""
    mode = _MODE if mode is None else mode
    L = _lib.lib()
    N, H, W, Cx = x.shape
    Cout = w.shape[0]
    assert w.shape[1] == Cx and Cx % 16 == 0, (tuple(w.shape), Cx)
    Hv, Wv = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = (Hv + 2 * (ks // 2) - ks) // stride + 1, (Wv + 2 * (ks // 2) - ks) // stride + 1
    out = th.empty((N, Ho, Wo, Cout), device=x.device, dtype=th.float32)
    rows = (Cout + 63) // 64 * 64
    scratch = th.empty(rows * Cx * ks * ks * 5 + 256 + (16 << 20) + N * Cx * H * W, device=x.device, dtype=th.float32)
    with _lib.on(x.device):
        _lib.check(L.hl_conv2d_nhwc_mode(mode, _lib.ptr(x), N, H, W, Cx, _lib.ptr(w), _lib.ptr(b), Cout, ks, stride, ups, None, None, 0,
                                         None, _lib.ptr(out), _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()), "hl_conv2d_nhwc_mode")
    return out


def _pad_c(x, mult):
    c = x.shape[-1]
    return x if c % mult == 0 else F.pad(x, (0, mult - c % mult))


class _Conv(th.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, ups):
        w4 = w.unsqueeze(-1) if w.dim() == 3 else w              # Conv1d k=1 (attention qkv / proj_out) is a 1x1 conv
        ks = w4.shape[2]
        Cin = w4.shape[1]
        wp = w4 if x.shape[-1] == Cin else F.pad(w4, (0, 0, 0, 0, 0, x.shape[-1] - Cin))    # zero weights for the padded input channels
        mode = _conv_mode()
        y = _conv_raw(x, wp.contiguous(), b, ks, stride, ups, mode)
        ctx.save_for_backward(x, w4)
        ctx.meta = (ks, stride, ups, w.shape, b is not None)
        ctx.mode = mode                                          # backward-data in the arithmetic of the forward (autocast is off in backward)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w4 = ctx.saved_tensors
        ks, stride, ups, wshape, has_b = ctx.meta
        L = _lib.lib()
        dy = dy.contiguous()
        # gradients do not fit fp16's range without loss scaling (values below 6e-8 vanish, 6e-5 and below lose bits): the backward of the
        # fp16 mode runs in bf16 - same kernels, fp32's exponent range - whether or not the caller wraps the step in a GradScaler
        bmode = _lib.HL_CONV_BF16 if ctx.mode == _lib.HL_CONV_FP16 else ctx.mode
        N, Ho, Wo, Cout = dy.shape
        Cin = w4.shape[1]
        dyp = _pad_c(dy, 16)                                      # (the 27-channel output conv: pad the gradient's channels with zeros)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # backward-data = forward conv kernels on dy with W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx], read in place from w4
            Cxp, Cyp = x.shape[-1], dyp.shape[-1]
            dx = th.empty_like(x) if Cxp == Cin else th.zeros_like(x)
            rows = (Cin + 63) // 64 * 64
            extra = N * 4 * Ho * Wo * Cyp if stride == 2 else (N * Ho * Wo * Cxp if ups else 0)
            scratch = th.empty(rows * Cyp * ks * ks * 5 + 512 + (16 << 20) + extra, device=dy.device, dtype=th.float32)
            with _lib.on(dy.device):
                _lib.check(L.hl_conv2d_nhwc_bwd_data(bmode, _lib.ptr(dyp), N, Ho, Wo, Cyp, _lib.ptr(w4.contiguous()), Cout, Cin, ks, stride, ups,
                                                     _lib.ptr(dx), Cxp, _lib.ptr(scratch), scratch.numel() * 4, _lib.stream_ptr()),
                           "hl_conv2d_nhwc_bwd_data")
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            dy2 = _pad_c(dy, 4)
            geom = (N, x.shape[1], x.shape[2], x.shape[3], dy2.shape[-1], ks, stride, ups, Cout, Cin)
            nbytes = L.hl_conv2d_wgrad_scratch_bytes(*geom)          # per-slab partial blocks (the kernels store every element of dw / db)
            part = th.empty(max(1, nbytes // 4), device=dy.device, dtype=th.float32)
            dw = th.empty(w4.shape, device=dy.device, dtype=th.float32)
            db = th.empty((Cout,), device=dy.device, dtype=th.float32) if has_b else None
            with _lib.on(dy.device):
                _lib.check(L.hl_conv2d_wgrad_nhwc_ws_mode(bmode, _lib.ptr(x), N, x.shape[1], x.shape[2], x.shape[3], _lib.ptr(dy2), dy2.shape[-1], ks,
                                                          stride, ups, _lib.ptr(dw), Cout, Cin, _lib.ptr(db), _lib.ptr(part), nbytes, _lib.stream_ptr()),
                           "hl_conv2d_wgrad_nhwc_ws_mode")
            dw = dw.reshape(wshape)
        return dx, dw, db, None, None


def conv(x, m, stride=1, ups=0):
    return _Conv.apply(x, m.weight, m.bias, stride, ups)


class _GroupNormAct(th.autograd.Function):
    """y = silu?( GroupNorm32(x) [* (1 + scale) + shift] )   (nn.py:17-19,100; unet.py:198-219, use_scale_shift_norm)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, ss, silu):
        L = _lib.lib()
        N, H, W, Cc = x.shape
        dev = x.device
        A, B = th.empty((N, Cc), device=dev), th.empty((N, Cc), device=dev)
        gstat = th.empty((N, 32, 2), device=dev)                      # (mean, rstd) per group
        y = th.empty((N, H, W, Cc), device=dev)
        scr = th.empty(N * 8192, device=dev)
        ssc = ss.contiguous() if ss is not None else None
        with _lib.on(dev):
            _lib.check(L.hl_groupnorm_train_forward(_lib.ptr(x), N, H, W, Cc, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(ssc), 1 if silu else 0,
                                                    _lib.ptr(A), _lib.ptr(B), _lib.ptr(gstat), _lib.ptr(y), _lib.ptr(scr), scr.numel() * 4,
                                                    _lib.stream_ptr()), "hl_groupnorm_train_forward")
        ctx.save_for_backward(x, A, B, gstat, gamma, beta, ssc if ssc is not None else th.empty(0, device=dev))
        ctx.silu = silu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, A, B, gstat, gamma, beta, ss = ctx.saved_tensors
        has_ss = ss.numel() > 0
        L = _lib.lib()
        N, H, W, Cc = x.shape
        dev = x.device
        dy = dy.contiguous()
        dx = th.empty_like(x) if ctx.needs_input_grad[0] else None
        dgamma, dbeta = th.empty(Cc, device=dev), th.empty(Cc, device=dev)
        dss = th.empty((N, 2 * Cc), device=dev) if has_ss else None
        scr = th.empty(N * Cc * 5 + L.hl_gn_backward_scratch_bytes(N, H * W, Cc) // 4, device=dev)
        with _lib.on(dev):
            _lib.check(L.hl_groupnorm_train_backward(_lib.ptr(x), _lib.ptr(dy), N, H, W, Cc, _lib.ptr(A), _lib.ptr(B), 1 if ctx.silu else 0,
                                                     _lib.ptr(gstat), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(ss) if has_ss else None, _lib.ptr(dx),
                                                     _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(dss), _lib.ptr(scr), scr.numel() * 4,
                                                     _lib.stream_ptr()), "hl_groupnorm_train_backward")
        return dx, dgamma, dbeta, dss, None


def gn_act(x, m, ss=None, silu=True):
    return _GroupNormAct.apply(x, m.weight, m.bias, ss, silu)


class _Attention(th.autograd.Function):
    """QKVAttention (unet.py:255-274) on qkv (N, T, 3C) with channel = head*3ch + {q|k|v}*ch + c -> (N, T, C)."""

    @staticmethod
    def forward(ctx, qkv, heads):
        N, T, C3 = qkv.shape
        Cc = C3 // 3
        out = th.empty((N, T, Cc), device=qkv.device)
        with _lib.on(qkv.device):
            _lib.check(_lib.lib().hl_attention_nhwc(_lib.ptr(qkv), N, T, Cc, heads, _lib.ptr(out), _lib.stream_ptr()), "hl_attention_nhwc")
        ctx.save_for_backward(qkv, out)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, do):
        # hl_attention_nhwc_backward (csrc/hl_attention_bwd.hip): flash-style fp32 on the matrix cores, probabilities recomputed, deterministic
        qkv, out = ctx.saved_tensors
        N, T, C3 = qkv.shape
        Cc = C3 // 3
        L = _lib.lib()
        do = do.contiguous().float()
        dqkv = th.empty_like(qkv)
        nbytes = L.hl_attention_backward_scratch_bytes(N, T, Cc, ctx.heads)
        scr = th.empty(nbytes // 4 + 16, device=qkv.device)
        with _lib.on(qkv.device):
            _lib.check(L.hl_attention_nhwc_backward(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(do), N, T, Cc, ctx.heads, _lib.ptr(dqkv), _lib.ptr(scr),
                                                    scr.numel() * 4, _lib.stream_ptr()), "hl_attention_nhwc_backward")
        return dqkv, None


# ---- the network (unet.py:550-615) on those ops ---------------------------------------------------------------------------------------
def _embedding(timesteps, dim):
    from .nn import timestep_embedding
    return timestep_embedding(timesteps, dim)                               # hl_timestep_embedding (no parameters)


def _silu(x):
    return x * th.sigmoid(x)


def _res_block(m, x, emb):
    h = conv(gn_act(x, m.in_layers[0]), m.in_layers[2])
    ss = m.emb_layers[1](_silu(emb))                                        # (N, 2*Cout): [scale | shift] (unet.py:203-206)
    if getattr(m, "use_scale_shift_norm", True):
        h = gn_act(h, m.out_layers[0], ss)
    else:                                                                   # (N, Cout) added before the GroupNorm (unet.py:216-218)
        h = gn_act(h + ss[:, None, None, :], m.out_layers[0], None)
    if m.dropout > 0:                                                       # out_layers[2] = nn.Dropout (unet.py:196): identity in eval mode
        h = F.dropout(h, p=m.dropout, training=m.training)
    h = conv(h, m.out_layers[3])
    skip = x if isinstance(m.skip_connection, th.nn.Identity) else conv(x, m.skip_connection)
    return skip + h


def _attention(m, x):
    N, H, W, Cc = x.shape
    qkv = conv(gn_act(x, m.norm, None, silu=False), m.qkv)                  # Conv1d k=1 == 1x1 conv
    a = _Attention.apply(qkv.reshape(N, H * W, 3 * Cc), m.num_heads)
    return x + conv(a.reshape(N, H, W, Cc), m.proj_out)


def _run(seq, h, emb):
    for m in seq:
        if isinstance(m, U.ResBlock):
            h = _res_block(m, h, emb)
        elif isinstance(m, U.AttentionBlock):
            h = _attention(m, h)
        elif isinstance(m, U.Downsample):
            h = conv(h, m.op, stride=2)
        elif isinstance(m, U.Upsample):
            h = conv(h, m.conv, ups=1)
        else:                                                               # the bare first convolution of an encoder
            h = conv(h, m)
    return h


def forward_train(model, x, timesteps, x_cond=None, y=None):
    """UNetModel.forward's contract (x (N,C,H,W), timesteps (N,), x_cond, y) -> (N,C_out,H,W), differentiable through HIP kernels."""
    if not x.is_cuda:
        raise RuntimeError("the HIP training path needs CUDA(HIP) tensors; there is no CPU path")
    if model.num_classes is not None:
        assert y is not None and y.shape == (x.shape[0],)
    if model.cond_type == "concat" and x_cond is not None:   # unet.py:572-573 (UNetModel.forward has already joined them when it is the caller)
        x, x_cond = th.cat([x, x_cond], dim=1), None
    if th.is_autocast_enabled():
        # The caller trains under autocast (train_util.py:214, --use_amp True).  The kernels in here take fp32 tensors, so torch's own
        # autocasting of the few tensor ops of this function is switched off; the convolutions take the autocast dtype as their operand
        # precision (k_conv_h16: 16-bit operands, fp32 accumulation) unless set_train_arithmetic pinned another arithmetic.
        dt = th.get_autocast_dtype("cuda") if hasattr(th, "get_autocast_dtype") else th.get_autocast_gpu_dtype()
        prev = getattr(_TLS, "autocast", None)
        _TLS.autocast = "bf16" if dt == th.bfloat16 else "fp16"
        if _ARITH["mode"] is None and _TLS.autocast not in _LOGGED:   # said once: autocast changes the arithmetic of the convolutions
            _LOGGED.add(_TLS.autocast)
            import warnings
            warnings.warn(f"humanliff_amd: training under torch.autocast - convolutions and 3x3 weight gradients take {_TLS.autocast} operands "
                          "(fp32 accumulation); set_train_arithmetic('fp32') pins fp32", stacklevel=3)
        try:
            with th.autocast(device_type="cuda", enabled=False):
                return forward_train(model, x.float(), timesteps, None if x_cond is None else x_cond.float(), y)
        finally:
            _TLS.autocast = prev
    emb = model.time_embed[2](_silu(model.time_embed[0](_embedding(timesteps, model.model_channels))))
    if model.cond_type == "AdaGN":           # unet.py:574-578 (like the embedding MLP: three tiny torch modules, autograd's own backward)
        assert x_cond is not None, "cond_type='AdaGN' needs x_cond"
        xp = model.conv_proj_2(model.conv_proj_1(x_cond.float()))
        emb = emb + model.linear(xp.reshape(xp.shape[0], -1))
    if model.num_classes is not None:
        emb = emb + model.label_emb(y)
    to_nhwc = lambda t: _pad_c(t.float().permute(0, 2, 3, 1), 16).contiguous()  # noqa: E731   (27 -> 32 channels, zeros)
    hs = []
    h = to_nhwc(x)
    for blk in model.input_blocks:
        h = _run(blk, h, emb)
        hs.append(h)
    h = _run(model.middle_block, h, emb)
    if model.cond_type == "controlnet":
        assert x_cond is not None, "cond_type='controlnet' needs x_cond (zeros for the first layer)"
        hs_cond = []
        hc = to_nhwc(x + x_cond)
        for blk, proj in zip(model.input_blocks_cond, model.input_blocks_proj_cond):
            hc = conv(_run(blk, hc, emb), proj)
            hs_cond.append(hc)
    for blk in model.output_blocks:
        skip = hs.pop()
        if model.cond_type == "controlnet":
            skip = skip + hs_cond.pop()
        h = _run(blk, th.cat([h, skip], dim=-1), emb)
    out = conv(gn_act(h, model.out[0]), model.out[2])                       # (N, H, W, C_out)
    out = out.permute(0, 3, 1, 2).contiguous().to(x.dtype)
    if out.requires_grad and model.training and model._any_param_requires_grad():
        # backward is what precedes an optimizer step: from here on the packed inference weights count as stale, also when a sampling
        # call between this forward and the step has re-packed them meanwhile (fused optimizers do not bump Tensor._version)
        out.register_hook(lambda g, m=weakref.ref(model): (setattr(m(), "_hip_stale", True) if m() is not None else None, g)[1])
    return out


class GraphedTrainStep:
    """The whole optimisation step of the reference's loop (train_util.py:200-246: training_losses on one microbatch -> loss.backward() ->
    optimizer.step()) captured ONCE into a HIP graph and replayed: every launch of the step is a kernel of this library or of torch on
    torch's current stream, nothing synchronises, so the ~1 300 launches a step enqueues from Python (70 ms of host time for 74 ms of
    kernels under bf16 autocast) become one graph launch - 75.2 -> 68.8 ms per step at microbatch 2 on one MI355X.  An extension: the
    reference has no counterpart; shapes, the optimizer's hyper-parameters and the conv arithmetic are frozen at capture.

        step = GraphedTrainStep(diffusion, model, opt, x_start, x_cond, t, {"y": y}, autocast=torch.bfloat16)
        for batch in data: loss = step(batch.x, batch.cond, t, {"y": batch.y})      # `loss` is the graph's output tensor

    optimizer: a capturable one (torch.optim.AdamW(..., fused=True, capturable=True)).  The three warm-up iterations torch asks for run on a
    side stream on the example batch; parameters and optimizer state are restored afterwards, so construction leaves the model as it
    was.  with_noise=True makes the q_sample noise an input of the step (step(..., noise=...)); otherwise training_losses draws it inside
    the graph (torch's graph-safe generator).  Gradients are left in .grad after every replay.  If eager steps ran before, drop their loss
    tensors first: a live loss keeps the parameters' AccumulateGrad nodes bound to the stream it was computed on, and capture (which has to
    run on a side stream) then fails inside torch."""

    def __init__(self, diffusion, model, optimizer, x_start, x_cond, t, model_kwargs=None, *, autocast=None, with_noise=False, warmup=3):
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("GraphedTrainStep needs a capturable optimizer, e.g. torch.optim.AdamW(params, fused=True, capturable=True)")
        if not isinstance(optimizer, (th.optim.Adam, th.optim.AdamW)):
            # the warm-up below resets a FRESH optimizer's state to zeros - right for Adam / AdamW (zero moments, step 0), wrong for
            # optimizers whose initial state is not zero (NAdam mu_product = 1, Rprop step_size = lr, ASGD eta / mu)
            raise ValueError("GraphedTrainStep supports torch.optim.Adam / AdamW (capturable=True)")
        self.model, self.optimizer = model, optimizer
        clone = lambda v: v.detach().clone() if th.is_tensor(v) else v  # noqa: E731
        self._x, self._t = clone(x_start), clone(t)
        self._xc = None if x_cond is None else clone(x_cond)
        self._kw = {k: clone(v) for k, v in (model_kwargs or {}).items()}
        self._noise = th.randn_like(self._x) if with_noise else None

        def body():
            with th.autocast(device_type="cuda", dtype=autocast, enabled=autocast is not None):
                losses = diffusion.training_losses(model, self._x, self._xc, self._t, model_kwargs=self._kw, noise=self._noise)
            loss = losses["loss"].mean()
            loss.backward()
            optimizer.step()
            return loss

        params = [p for g in optimizer.param_groups for p in g["params"]]
        saved_p = [p.detach().clone() for p in params]
        tensors = lambda p: {k: v.detach().clone() for k, v in optimizer.state[p].items() if th.is_tensor(v)}  # noqa: E731
        fresh = not any(len(optimizer.state.get(p, {})) for p in params)
        saved_s = None if fresh else [tensors(p) for p in params]
        # the warm-up must leave no trace in the device RNG stream (dropout, q_sample noise): the eager run that did not warm up draws the same numbers.
        # (.grad belongs to the graph from here on: gradients accumulated before construction are discarded - step() leaves each replay's in place)
        rng_state = th.cuda.get_rng_state(self._x.device)
        side = th.cuda.Stream(device=self._x.device)
        side.wait_stream(th.cuda.current_stream(self._x.device))
        with th.cuda.stream(side):
            optimizer.zero_grad(set_to_none=True)
            body()                                           # (a fresh optimizer creates its state here)
            if fresh:                                        # the state before the first step: zero moments, step 0
                saved_s = [tensors(p) for p in params]
                for s_ in saved_s:
                    for v in s_.values():
                        v.zero_()
            for _ in range(max(0, warmup - 1)):
                optimizer.zero_grad(set_to_none=True)
                body()
            with th.no_grad():
                for p, q in zip(params, saved_p):
                    p.copy_(q)
                for p, s_ in zip(params, saved_s):
                    for k, v in s_.items():
                        optimizer.state[p][k].copy_(v)
        th.cuda.current_stream(self._x.device).wait_stream(side)
        th.cuda.set_rng_state(rng_state, self._x.device)
        self.graph = th.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with th.cuda.graph(self.graph):
            self.loss = body().detach()
        self._mark_stale()

    def _mark_stale(self):          # what forward_train's backward hook does in eager mode: the packed inference weights are out of date
        for m in self.model.modules():
            if hasattr(m, "_hip"):
                m._hip_stale = True

    def __call__(self, x_start, x_cond, t, model_kwargs=None, noise=None):
        self._x.copy_(x_start)
        self._t.copy_(t)
        if self._xc is not None:
            self._xc.copy_(x_cond)
        for k, v in (model_kwargs or {}).items():
            if th.is_tensor(self._kw.get(k)):
                self._kw[k].copy_(v)
        if self._noise is not None:
            if noise is None:
                raise ValueError("this step was captured with_noise=True: pass noise=")
            self._noise.copy_(noise)
        self.graph.replay()
        self._mark_stale()
        return self.loss
