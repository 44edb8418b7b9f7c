"""GPU tests at BASELINE.json's full sizes: the production 497M-parameter UNet against the oracle, and
size-independent properties of the renderer / sampler at 512x512 x (128+128) and 4x27x256x256."""
import pytest
import numpy as np
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


@pytest.fixture(scope="module")
def production():
    import bench
    model, diffusion, sd = bench.build_unet(dev)
    return model, diffusion, sd


def test_production_unet_matches_oracle(production):
    """F4 config (256x256x27, 192 base channels, 3 res blocks, attention at 32/16/8, controlnet, class-cond),
    one sample: HIP forward vs the CPU oracle on identical seeded weights/inputs."""
    from oracle import unet_oracle as uo
    model, _, sd = production
    g = torch.Generator().manual_seed(123)
    x = torch.randn((1, 27, 256, 256), generator=g)
    xc = torch.randn((1, 27, 256, 256), generator=g).clamp(-1, 1) * 0.7
    t = torch.tensor([617])
    y = torch.tensor([2])
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        want = uo.unet_forward(sd, x, t, xc, y, num_heads=4)
        got = model(x.to(dev), t.to(dev), xc.to(dev), y=y.to(dev)).cpu()
    scale = float(want.abs().mean())
    err = float((got - want).abs().max())
    assert scale > 0.05                      # a vacuous all-zero output would not count
    assert err < 5e-5 * max(1.0, scale), (err, scale)          # fp32, 256 stacked convs with K up to 13824; measured 4.5e-6 on outputs O(0.5)
    mse = float(((got - want) ** 2).mean())
    assert mse < 1e-9 * max(1.0, scale ** 2)
    # the other arithmetic modes meet the same bounds against the oracle: direct-only fp32 (no Winograd) and the opt-in
    # bf16x3 emulation of the fp32 products
    errs = {"fp32": err}
    for mode in ("fp32_mfma", "fp32_direct", "bf16x3"):      # ("fp32_mfma": the default's dispatch with every product on the fp32 matrix pipe, no fp16x2 kernels)
        model.set_conv_mode(mode)
        try:
            with torch.no_grad():
                alt = model(x.to(dev), t.to(dev), xc.to(dev), y=y.to(dev)).cpu()
            if mode == "fp32_mfma":
                assert sum(model.dispatch_census()["fp16x2"]) == 0 and sum(model.dispatch_census()["wino4"]) > 0
        finally:
            model.set_conv_mode("fp32")
        errs[mode] = float((alt - want).abs().max())
        assert not torch.equal(alt, got)     # the mode really switched kernels
        assert errs[mode] < 5e-5 * max(1.0, scale), (mode, errs, scale)      # measured 4.5e-6 / 5.0e-6 / 5.1e-6
        assert float(((alt - want) ** 2).mean()) < 1e-9 * max(1.0, scale ** 2)
    print("production UNet max-abs vs oracle: " + ", ".join(f"{k} {v:.3e}" for k, v in errs.items()) + f" (output scale {scale:.3f})")
    # the opt-in fp16-operand mode (k_conv_h16 on the 3x3 / stride-1 layers, fp32 accumulation, everything else fp32): the operand precision
    # of the reference's own TF32 convolutions - not an fp32-tolerance mode, so its bound is a PSNR
    model.set_conv_mode("fp16")
    try:
        with torch.no_grad():
            alt = model(x.to(dev), t.to(dev), xc.to(dev), y=y.to(dev)).cpu()
        census = model.dispatch_census()
    finally:
        model.set_conv_mode("fp32")
    assert sum(census["bf16x3"]) > 20, census                  # (the census counts k_conv_h16 with the kernels of the 16-bit matrix pipe)
    peak = float(want.abs().max())
    psnr = 10 * np.log10(peak ** 2 / float(((alt - want) ** 2).mean()))
    print(f"fp16-operand mode: max-abs {float((alt - want).abs().max()):.3e}, PSNR vs oracle {psnr:.1f} dB, 16-bit launches per level {census['bf16x3'][:6]}")
    assert psnr > 60, psnr                                      # measured ~75 dB


def test_production_b4_dispatch_matches_oracle(production):
    """The HEADLINE configuration's dispatch (BASELINE configs[1]: batch 4).  Kernel selection depends on the batch size - a 3x3 layer
    takes a Winograd F(4x4,3x3) kernel only where its workgroups fill the chip (which of the two kernels, and into how many slabs the
    input channels are split, depends on the number of 32x16-pixel tiles = on the batch size) - so the B=1 test above does not cover what
    the bench measures.  Here: one B=4 forward and four recurrent p_sample steps (t = 999..996, different noise per sample) of the production net,
    HIP vs the CPU oracle on identical x_T / noise, with the dispatch census asserting which kernels ran."""
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    from oracle import diffusion_oracle as do
    from oracle import unet_oracle as uo
    model, _, sd = production
    B, n_steps = 4, 4
    g = torch.Generator().manual_seed(321)
    x = x_T = torch.randn((B, 27, 256, 256), generator=g)
    xc = torch.zeros_like(x)
    xc[1:] = torch.randn((B - 1, 27, 256, 256), generator=g).clamp(-1, 1) * 0.5      # sample 0: first cloth layer (zeros), the others conditioned
    y = torch.tensor([0, 1, 2, 3])
    noises = [torch.randn(x.shape, generator=g) for _ in range(n_steps)]
    s = do.Schedule(do.linear_betas(1000), list(range(1000)))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    eps0 = None
    with torch.no_grad():
        for i in range(n_steps):
            t = torch.full((B,), 999 - i)
            eps = uo.unet_forward(sd, x, t, xc, y, num_heads=4)
            if i == 0:
                eps0 = eps
            x, _ = do.p_sample_step(s, x, t, eps, noises[i])
    # ---- HIP: the forward alone, then the recurrent steps through GaussianDiffusion.p_sample ----
    with torch.no_grad():
        got0 = model(x_T.to(dev), torch.full((B,), 999, device=dev), xc.to(dev), y=y.to(dev)).cpu()
    census = model.dispatch_census()
    print("dispatch at B=4:", {k: v[:6] for k, v in census.items() if any(v)})
    # the headline dispatch (late round 5): every 3x3 / stride-1 and 1x1 layer of the 256- ... 16-pixel levels on the direct fp16x2 kernels (k_conv_h2s,
    # k_conv1_h2s: two workgroups per CU; the 32- and 16-pixel levels with their input channels split into slabs); only the 27-channel input convolution stays on
    # F(4x4,3x3); no bf16 emulation in the default mode
    assert sum(census["wino4"]) <= 1 and sum(census["wino2"]) == 0 and census["bf16x3"] == [0] * 8, census
    assert all(census["fp16x2"][l] > 20 for l in range(5)), census
    scale = float(eps0.abs().mean())
    e0 = float((got0 - eps0).abs().max())
    assert scale > 0.05 and e0 < 5e-5 * max(1.0, scale), (e0, scale)        # measured ~5e-6, like B=1
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="")
    k = {"i": 0}
    orig = torch.randn_like

    def inj(ref):
        k["i"] += 1
        return noises[k["i"] - 1].to(ref.device)
    torch.randn_like = inj
    try:
        xg = x_T.to(dev)
        with torch.no_grad():
            for i in range(n_steps):
                xg = d.p_sample(model, xg, xc.to(dev), torch.full((B,), 999 - i, device=dev), model_kwargs={"y": y.to(dev)})["sample"]
    finally:
        torch.randn_like = orig
    assert k["i"] == n_steps
    err = float((xg.cpu() - x).abs().max())
    print(f"B=4 forward max-abs vs oracle {e0:.3e} (scale {scale:.3f}); x after {n_steps} recurrent p_sample steps max-abs {err:.3e} (|x| max {float(x.abs().max()):.2f})")
    assert err < 1e-4, err                                                  # values up to ~5; the B=1 figure of the bench line is ~5e-6
    # ... and the B=1 dispatch really is a different one (what makes this test necessary)
    with torch.no_grad():
        model(x_T[:1].to(dev), torch.full((1,), 999, device=dev), xc[:1].to(dev), y=y[:1].to(dev))
    c1 = model.dispatch_census()
    print("dispatch at B=1:", {k: v[:6] for k, v in c1.items() if any(v)})
    assert c1 != census, (c1, census)


def test_production_b8_dispatch_matches_oracle(production):
    """The batch the configs[3] / [4] slice really samples with (8 subjects per GPU at a time, triplane_sample_layered.py:112-134 with
    --batch_size 8): kernel selection depends on the batch size, so B = 1 and B = 4 above do not cover it.  One B = 8 forward of the
    production net, distinct y / x_cond / x per sample, HIP vs the CPU oracle, with the dispatch census asserted."""
    from oracle import unet_oracle as uo
    model, _, sd = production
    B = 8
    g = torch.Generator().manual_seed(808)
    x = torch.randn((B, 27, 256, 256), generator=g)
    xc = torch.zeros_like(x)
    xc[2:] = torch.randn((B - 2, 27, 256, 256), generator=g).clamp(-1, 1) * 0.5      # samples 0, 1: first cloth layer (zeros), the others conditioned
    y = torch.tensor([0, 0, 1, 2, 3, 1, 2, 3])
    t = torch.tensor([999, 617, 400, 999, 20, 0, 777, 250])
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        want = torch.cat([uo.unet_forward(sd, x[i:i + 2], t[i:i + 2], xc[i:i + 2], y[i:i + 2], num_heads=4) for i in range(0, B, 2)])
        got = model(x.to(dev), t.to(dev), xc.to(dev), y=y.to(dev)).cpu()
    census = model.dispatch_census()
    print("dispatch at B=8:", {k: v[:6] for k, v in census.items() if any(v)})
    assert all(census["fp16x2"][l] > 20 for l in range(4)) and sum(census["wino4"][:3]) <= 1, census   # the 256- ... 32-pixel levels on the direct fp16x2 kernels at this batch
    assert census["bf16x3"] == [0] * 8, census                              # no 16-bit emulation in the default mode
    scale = float(want.abs().mean())
    errs = [float((got[i] - want[i]).abs().max()) for i in range(B)]
    print(f"B=8 forward max-abs vs oracle per sample {['%.2e' % e for e in errs]} (scale {scale:.3f})")
    assert scale > 0.05 and max(errs) < 5e-5 * max(1.0, scale), (errs, scale)   # the B = 1 / B = 4 bound
    # (per kernel family and level the census equals the B = 4 one; inside a family the batch size picks the kernel variant - 64- or
    #  32-channel F(4x4) workgroups - and the number of input-channel slabs, which is why this batch has its own oracle check)


def test_production_unet_with_small_zero_module_weights_matches_oracle():
    """Round 6: the weights a TRAINED checkpoint has and the seeded synthetic one does not.  Every zero_module convolution of the reference
    (unet.py:149 out_layers[-1] of each ResBlock, :237 proj_out of each attention, :462 out[-1], :494-518 the 24 control zero-convolutions) starts at 0 and stays
    small early in training: here N(0, 1e-4^2) for weights and biases, the GroupNorm affines jittered as in every fixture.  Unscaled fp16 weight planes (round 5)
    lose their low plane entirely at that magnitude (TF32's precision); with the per-channel power-of-two scaling the default mode meets the same RELATIVE bound as
    the all-fp32-MFMA mode.  B = 1 and B = 2 (another dispatch), outputs O(1e-4)."""
    import bench
    from humanliff_amd import synthetic as syn
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion
    from oracle import unet_oracle as uo
    model, _ = create_model_and_diffusion(**bench.F4)
    keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    sd = syn.state_from_shapes(keys, seed=4)
    gz = torch.Generator().manual_seed(99)
    n_small = 0
    for k in sd:
        stem = k.rsplit(".", 1)[0]
        if stem.endswith("out_layers.3") or stem.endswith("proj_out") or stem == "out.2" or stem.startswith("input_blocks_proj_cond."):
            sd[k] = torch.randn(sd[k].shape, generator=gz) * 1e-4
            n_small += 1
    assert n_small == 2 * (62 + 31 + 1 + 24), n_small          # ResBlocks, attention blocks, the output convolution, the control projections
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(124)
    B = 2
    x = torch.randn((B, 27, 256, 256), generator=g)
    xc = torch.randn((B, 27, 256, 256), generator=g).clamp(-1, 1) * 0.7
    t, y = torch.tensor([617, 40]), torch.tensor([2, 0])
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        want = uo.unet_forward(sd, x, t, xc, y, num_heads=4)
    scale = float(want.abs().mean())
    assert 1e-6 < scale < 1e-2, scale                           # the output convolution is one of the small ones
    res = {}
    for mode in ("fp32", "fp32_mfma"):
        model.set_conv_mode(mode)
        with torch.no_grad():
            got2 = model(x.to(dev), t.to(dev), xc.to(dev), y=y.to(dev)).cpu()
            got1 = model(x[:1].to(dev), t[:1].to(dev), xc[:1].to(dev), y=y[:1].to(dev)).cpu()
        if mode == "fp32":
            assert sum(model.dispatch_census()["fp16x2"]) > 100
        res[mode] = (float((got2 - want).abs().max()) / scale, float((got1 - want[:1]).abs().max()) / scale,
                     float((got2 - want).norm() / want.norm()))
    model.set_conv_mode("fp32")
    print(f"small zero-module weights: output mean-abs {scale:.3e}; (max-abs / mean-abs at B=2, at B=1, rel-L2): fp16x2 default {res['fp32']}, all-fp32-MFMA {res['fp32_mfma']}")
    for mode in res:
        assert res[mode][0] < 1e-4 and res[mode][1] < 1e-4 and res[mode][2] < 5e-6, (mode, res)      # (5e-5 of an O(0.5) output in the other production tests is the same 1e-4 of the mean-abs)
    assert res["fp32"][2] < 2.0 * res["fp32_mfma"][2] + 1e-7, res


@pytest.mark.parametrize("image_size,B", [(128, 3), (64, 5)])
def test_other_image_sizes_dispatch_matches_oracle(image_size, B):
    """The kernel selection of conv2d is a function of (pixels, channels, batch) with a dozen thresholds; the census tests above pin it at 256 x 256 only.  Here the
    production widths (192 base channels, controlnet, class-cond, scale-shift, 4 heads) at 128 x 128 (channel_mult (1, 1, 2, 3, 4): 576-channel levels, a width the
    256-pixel net does not have) and 64 x 64 ((1, 2, 3, 4)), odd batch sizes: every kernel family the dispatch picks there is checked against the oracle, and the
    census is printed and bounded so that a threshold change that re-routes these sizes shows up."""
    import bench
    from humanliff_amd import synthetic as syn
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion
    from oracle import unet_oracle as uo
    cfg = dict(bench.F4, image_size=image_size, num_res_blocks=2, attention_resolutions="16,8")
    model, _ = create_model_and_diffusion(**cfg)
    keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    sd = syn.state_from_shapes(keys, seed=9)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(image_size + B)
    x = torch.randn((B, 27, image_size, image_size), generator=g)
    xc = torch.randn((B, 27, image_size, image_size), generator=g).clamp(-1, 1) * 0.6
    t = torch.randint(0, 1000, (B,), generator=g)
    y = torch.randint(0, 4, (B,), generator=g)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        want = uo.unet_forward(sd, x, t, xc, y, num_heads=4)
        got = model(x.to(dev), t.to(dev), xc.to(dev), y=y.to(dev)).cpu()
    census = model.dispatch_census()
    scale, err = float(want.abs().mean()), float((got - want).abs().max())
    print(f"{image_size} x {image_size}, B = {B}: max-abs vs oracle {err:.3e} (scale {scale:.3f}); dispatch", {k: v[:6] for k, v in census.items() if any(v)})
    assert scale > 0.05 and err < 5e-5 * max(1.0, scale), (err, scale)
    assert sum(census["fp16x2"]) >= 30 and census["bf16x3"] == [0] * 8, census        # the default mode's direct fp16x2 kernels carry these sizes too; no 16-bit emulation
    assert sum(census["direct"]) + sum(census["wino2"]) + sum(census["wino4"]) >= 1, census
    # ... and the all-fp32 dispatch of the same sizes (the Winograd / direct kernels the fp16x2 ones replaced)
    model.set_conv_mode("fp32_mfma")
    try:
        with torch.no_grad():
            alt = model(x.to(dev), t.to(dev), xc.to(dev), y=y.to(dev)).cpu()
        c32 = model.dispatch_census()
    finally:
        model.set_conv_mode("fp32")
    e32 = float((alt - want).abs().max())
    print(f"    fp32_mfma mode: {e32:.3e}; dispatch", {k: v[:6] for k, v in c32.items() if any(v)})
    assert sum(c32["fp16x2"]) == 0 and e32 < 5e-5 * max(1.0, scale), (e32, c32)


def test_attention_branch_with_tiny_values_matches_oracle():
    """Round 6 ("the attention's K/V side"): production widths at 64 x 64 with attention on the 32- and 8-pixel levels (heads of 96 and 192 channels: the
    fp16x2 key-split kernels), every attention block's V rows of the qkv convolution (weights and bias) x 2^-14 and its proj_out weights x 2^14 - the branch contributes at its
    usual magnitude, but V and the attention's output are 6e-5 of it.  Unscaled fp16 planes lose the low plane of both there (attention alone: rel-L2 2.8e-4 against float64);
    with the running V scale inside the attention and the projection convolution's activation scale from the attention's own sum x^2 the usual bound holds."""
    import bench
    from humanliff_amd import synthetic as syn
    from humanliff_amd.improved_diffusion.script_util import create_model_and_diffusion
    from oracle import unet_oracle as uo
    cfg = dict(bench.F4, image_size=64, num_res_blocks=1, attention_resolutions="32,8")
    model, _ = create_model_and_diffusion(**cfg)
    keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    sd = syn.state_from_shapes(keys, seed=21)
    heads, n_att, widths = 4, 0, set()
    for k in list(sd):
        if k.endswith(".qkv.weight"):
            p_ = k[:-len(".qkv.weight")]
            C = sd[k].shape[1]
            ch = C // heads
            rows = torch.cat([torch.arange(h * 3 * ch + 2 * ch, h * 3 * ch + 3 * ch) for h in range(heads)])
            sd[k][rows] *= 2.0 ** -14
            sd[p_ + ".qkv.bias"][rows] *= 2.0 ** -14
            sd[p_ + ".proj_out.weight"] *= 2.0 ** 14
            n_att += 1
            widths.add(ch)
    assert n_att >= 4 and widths == {96, 192}, (n_att, widths)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(77)
    B = 2
    x = torch.randn((B, 27, 64, 64), generator=g)
    xc = torch.randn((B, 27, 64, 64), generator=g).clamp(-1, 1) * 0.6
    t, y = torch.tensor([801, 17]), torch.tensor([1, 3])
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        want = uo.unet_forward(sd, x, t, xc, y, num_heads=heads)
        got = model(x.to(dev), t.to(dev), xc.to(dev), y=y.to(dev)).cpu()
        model.set_conv_mode("fp32_mfma")
        try:
            alt = model(x.to(dev), t.to(dev), xc.to(dev), y=y.to(dev)).cpu()
        finally:
            model.set_conv_mode("fp32")
    scale = float(want.abs().mean())
    err, e32 = float((got - want).abs().max()), float((alt - want).abs().max())
    print(f"tiny attention values: max-abs vs oracle, default {err:.3e}, all-fp32-MFMA {e32:.3e} (output mean-abs {scale:.3f})")
    assert scale > 0.05 and err < 5e-5 * max(1.0, scale) and e32 < 5e-5 * max(1.0, scale), (err, e32, scale)
    assert err < 3.0 * e32 + 1e-6, (err, e32)


def test_configs1_full_length_loop_is_finite_bounded_and_reproducible(production):
    """BASELINE configs[1] at its real length: 1000-step p_sample_loop of the production net at B = 4 (x_cond = zeros, y = zeros, clip_denoised) - what the bench times for 20
    steps.  No reference trajectory of that length exists (5 - 6 h of CPU), so: every value finite, the final sample inside [-1, 1] (the last step returns the clamped
    x_0 estimate: posterior_mean_coef2[0] = 0), not collapsed, and a second run from the same generator state bit-identical (every kernel of the forward sums in a
    fixed order; GroupNorm totals are integer).  ~27 s per run."""
    model, diffusion, _ = production

    def run():
        torch.manual_seed(1234)
        g = torch.Generator(device=dev).manual_seed(77)
        noise = torch.randn((4, 27, 256, 256), device=dev, generator=g)
        xc = torch.zeros_like(noise)
        y = torch.zeros((4,), dtype=torch.int64, device=dev)
        with torch.no_grad():
            return diffusion.p_sample_loop(model, (4, 27, 256, 256), x_cond=xc, noise=noise, clip_denoised=True, model_kwargs={"y": y}, device=dev)
    a = run()
    assert a.shape == (4, 27, 256, 256) and bool(torch.isfinite(a).all())
    assert float(a.abs().max()) <= 1.0 and float(a.std()) > 1e-3, (float(a.abs().max()), float(a.std()))
    b = run()
    assert torch.equal(a, b)
    print(f"1000-step p_sample_loop at B = 4: |x| max {float(a.abs().max()):.4f}, std {float(a.std()):.4f}, two runs bit-identical")


def test_subject_sampled_in_a_batch_of_8_equals_the_same_subject_alone(production):
    """The e2e slice samples 8 subjects at a time; its oracle check is of the renderer on the generated tri-plane, so this ties the
    8-at-a-time sampler to the B = 1 sampler (which IS pinned to the reference: chain_f4_ddim10.npz, f4_ddim50.npz, f4_p250.npz): 4 cloth
    layers x DDIM-10 chained through x_cond (triplane_sample_layered.py:112-134), per-subject x_T seeds; subjects 0, 3 and 7 sampled
    inside the batch of 8 equal the same subjects sampled alone within the bound of the reference chain test."""
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    model, _, _ = production
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="ddim10")
    B, layers = 8, 4

    def x_T(k, layer):
        return torch.randn((27, 256, 256), generator=torch.Generator().manual_seed(5000 + 10 * k + layer))

    def chain(ids):
        xc = torch.zeros((len(ids), 27, 256, 256), device=dev)
        outs = []
        with torch.no_grad():
            for layer in range(layers):
                noise = torch.stack([x_T(k, layer) for k in ids]).to(dev)
                yy = torch.full((len(ids),), layer, dtype=torch.int64, device=dev)
                xc = d.ddim_sample_loop(model, noise.shape, x_cond=xc, noise=noise, clip_denoised=True, model_kwargs={"y": yy}, device=dev)
                outs.append(xc.cpu())
        return outs
    batch = chain(list(range(B)))
    census8 = model.dispatch_census()
    for k in (0, 3, 7):
        alone = chain([k])
        for layer in range(layers):
            a, b = batch[layer][k], alone[layer][0]
            err = float((a - b).abs().max())
            print(f"subject {k} layer {layer}: batch-of-8 vs alone max-abs {err:.3e}")
            assert torch.isfinite(a).all() and float(a.abs().max()) > 0.5
            # DDIM (eta = 0) re-injects nothing: the ~1e-5 per-forward difference between the two kernel selections grows over the 10 network
            # evaluations of a layer and rides on through x_cond.  Measured on MI355X: 1.9e-4 ... 2.7e-4 after layer 0, 7.2e-4 ... 9.7e-4 after
            # layer 3 (values in [-1, 1]; mean-abs below 1e-5).  The reference chain test allows 1e-3 per layer against the reference.
            assert err < 1e-3 * (layer + 1), (k, layer, err)
            assert float((a - b).abs().mean()) < 1e-5 * (layer + 1)
            assert 10 * np.log10(1.0 / float(((a - b) ** 2).mean())) > 90.0
    assert model.dispatch_census() != census8                              # the two really took different kernel selections


def test_production_ddim50_matches_reference(production):
    """DDIM-50 on the production network - the sampler length of BASELINE configs[3] / [4] and of the shipped sampling scripts - against
    the reference's own trajectory on identical noise (tests/golden/f4_ddim50.npz): after steps 1, 10, 25, 40 and 50."""
    import bench
    model, _, _ = production
    res = bench.ddim50_parity(model, dev)
    assert res["ndraws"] == res["ndraws_reference"] == 51                  # x_T + one randn_like per step (drawn although eta = 0)
    assert [s["step"] for s in res["steps"]] == [1, 10, 25, 40, 50]
    for s in res["steps"]:
        print(s)
        # DDIM with eta = 0 re-injects nothing, so differences accumulate over the 50 network evaluations; values reach 5.3 mid-loop and
        # [-1, 1] at the end.  Measured on MI355X: max-abs 7.7e-7 / 1.8e-6 / 9.0e-6 / 2.6e-5 / 3.0e-5 after steps 1 / 10 / 25 / 40 / 50
        # (PSNR 142 ... 118.7 dB), |sum| relative 2e-8 ... 1.6e-7 in round 5 (unscaled fp16x2 planes; 2.6e-5 with every product on the fp32 pipe).  Round 6, scale-invariant
        # planes: 4.2e-7 / 1.4e-6 / 5.8e-6 / 1.5e-5 / 1.8e-5 (142 ... 122.5 dB).  The bound is VERDICT r05's "done" figure for the default mode, 4e-5 at every step.
        assert s["max_abs"] < 4e-5, s
        assert s["psnr_db"] > 115.0 and s["abs_sum_rel"] < 1e-6, s
    assert res["final_row100_max_abs"] < 4e-5 and res["final_channel_mean_max_abs"] < 5e-7


def test_production_shipped_sampler_p250_matches_reference(production):
    """The SHIPPED sampler configuration (triplane_scripts/SynBody_triplane_sample_layered_*.sh:24-26: --timestep_respacing 250, p_sample_loop,
    --batch_size 1) on the production network against the reference's own trajectory on identical noise (tests/golden/f4_p250.npz): after
    steps 1, 50, 125, 200 and 250.  Ancestral sampling re-injects noise at every step and contracts differences (shown on the small nets in
    test_unet_gpu.py), so the bound of the DDIM-50 test holds with room."""
    import os
    import bench
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f4_p250.npz")):
        pytest.skip("tests/golden/f4_p250.npz not generated (tests/golden/gen_golden_p250.py, 25 min of CPU in the build container)")
    model, _, _ = production
    res = bench.ddim50_parity(model, dev, kind="p250")
    assert res["ndraws"] == res["ndraws_reference"] == 251
    assert [s["step"] for s in res["steps"]] == [1, 50, 125, 200, 250]
    for s in res["steps"]:
        print(s)
        assert s["max_abs"] < 3e-4 and s["psnr_db"] > 105.0 and s["abs_sum_rel"] < 2e-6, s
    assert res["final_row100_max_abs"] < 3e-4 and res["final_channel_mean_max_abs"] < 1e-6


def test_production_fp16_conv_mode_ddim50_against_the_reference(production):
    """The OPT-IN 16-bit mode of the UNet (set_conv_mode('fp16'): fp16 operands / fp32 accumulation in the 3x3 and 1x1 convolutions) pinned to
    the REFERENCE's DDIM-50 trajectory - not to this library's own fp32 run: a floor on what the mode delivers after 50 recurrent
    evaluations (values in [-1, 1] at the end, up to 5.3 mid-loop).  Not an fp32-tolerance mode; never the headline."""
    import bench
    model, _, _ = production
    model.set_conv_mode("fp16")
    try:
        res = bench.ddim50_parity(model, dev)
    finally:
        model.set_conv_mode("fp32")
    for s in res["steps"]:
        print(s)
    assert res["ndraws"] == res["ndraws_reference"] == 51
    assert res["steps"][0]["psnr_db"] > 70.0                       # one evaluation
    assert res["psnr_db"] > 55.0 and res["steps"][-1]["max_abs"] < 0.05   # measured ~60 dB after 50 steps (round 3, against the fp32 run)
    assert res["final_channel_mean_max_abs"] < 1e-3


def test_production_batch_independence(production):
    """Samples of a batch do not interact (GroupNorm/attention are per sample): B=2 == two B=1 calls, up to the
    summation order (the split-K factor of the low-resolution layers depends on the batch size)."""
    model, _, _ = production
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 27, 256, 256), generator=g).to(dev)
    xc = torch.zeros_like(x)
    t = torch.tensor([900, 30], device=dev)
    y = torch.tensor([0, 3], device=dev)
    both = model(x, t, xc, y=y)
    one0 = model(x[:1], t[:1], xc[:1], y=y[:1])
    one1 = model(x[1:], t[1:], xc[1:], y=y[1:])
    assert float((both[0] - one0[0]).abs().max()) < 2e-4 and float((both[1] - one1[0]).abs().max()) < 2e-4
    # while the same call twice is bit-identical (no atomics anywhere in the path)
    assert torch.equal(both, model(x, t, xc, y=y))


def test_sampler_update_full_size_vs_oracle():
    from humanliff_amd.improved_diffusion.script_util import create_gaussian_diffusion
    from oracle import diffusion_oracle as do
    d = create_gaussian_diffusion(steps=1000, timestep_respacing="")
    s = do.Schedule(do.linear_betas(1000), list(range(1000)))
    g = torch.Generator().manual_seed(9)
    shape = (4, 27, 256, 256)
    x, eps, noise = [torch.randn(shape, generator=g) for _ in range(3)]
    t = torch.tensor([999, 500, 1, 0])
    ps, x0 = d._step(0, x.to(dev), eps.to(dev), noise.to(dev), t.to(dev), True)
    want, want0 = do.p_sample_step(s, x, t, eps, noise, True)
    near = lambda a, b: bool(((a.cpu() - b).abs() <= 3e-7 * b.abs() + 2e-7).all())  # noqa: E731
    assert near(ps, want) and near(x0, want0)
    dd, _ = d._step(1, x.to(dev), eps.to(dev), noise.to(dev), t.to(dev), True, eta=0.3)
    wantd, _ = do.ddim_step(s, x, t, eps, noise, True, 0.3)
    assert near(dd, wantd)


@pytest.fixture(scope="module")
def view512():
    from humanliff_amd import synthetic as syn
    from humanliff_amd.NeRF import Renderer
    planes = syn.triplane(seed=11).to(dev)
    mlp = syn.render_mlp_state(3, gain=2.0)
    r = Renderer(use_canonical_space=False, triplane_dim=256, triplane_ch=27, test=True)
    r.load_state_dict(mlp, strict=False)
    r = r.to(dev)
    rays = [t.to(dev) for t in syn.orbit_rays(5, 36, 512, 512)]
    u = torch.rand((512 * 512, 128), generator=torch.Generator().manual_seed(5)).to(dev)
    tp = {"world_bounds": torch.tensor(syn.WORLD_BOUNDS)[None].to(dev)}

    def run(idx=None):
        ro, rd, nr, fr = [t if idx is None else t[idx] for t in rays]
        uu = u if idx is None else u[idx]
        out = r.render(tp, None, None, ro[None], rd[None], nr[None], fr[None], planes, 128, False, n_samples=128, u=uu)
        return {k: v[0].clone() for k, v in out.items()}
    return run, planes, mlp, rays, u


def test_render_512_properties(view512):
    run, *_ = view512
    full = run()
    assert full["rgb_map"].shape == (512 * 512, 3)
    assert torch.isfinite(full["rgb_map"]).all()
    # last section has alpha = 1, so the weights telescope to 1 (renderer.py:185-186, 221)
    assert float((full["acc_map"] - 1).abs().max()) < 1e-3
    assert float(full["depth_map"].min()) >= 0 and float(full["depth_map"].max()) <= 1
    assert float(full["rgb_map"].min()) >= 0 and float(full["rgb_map"].max()) <= 1 + 1e-3
    # rays are independent: any subset / permutation renders to bit-identical values
    perm = torch.randperm(512 * 512, generator=torch.Generator().manual_seed(1)).to(dev)
    shuf = run(perm)
    assert torch.equal(shuf["rgb_map"], full["rgb_map"][perm])
    assert torch.equal(shuf["depth_map"], full["depth_map"][perm])
    part = run(torch.arange(1000, 1000 + 70001, device=dev))          # ragged count
    assert torch.equal(part["rgb_map"], full["rgb_map"][1000:1000 + 70001])
    # deterministic
    again = run()
    assert torch.equal(again["rgb_map"], full["rgb_map"])


def test_render_512_sample_vs_oracle(view512):
    from humanliff_amd import synthetic as syn
    from oracle import render_oracle as ro
    run, planes, mlp, rays, u = view512
    idx = torch.randperm(512 * 512, generator=torch.Generator().manual_seed(2))[:3000]
    got = run(idx.to(dev))
    o, d, nr, fr = [t[idx.to(dev)].cpu() for t in rays]
    rgb, acc, depth = ro.render_rays(mlp, planes[0].cpu(), torch.tensor(syn.WORLD_BOUNDS), o, d, nr, fr, 128, 128,
                                     u=u[idx.to(dev)].cpu())
    assert float((got["rgb_map"].cpu() - rgb).abs().max()) < 5e-5
    assert float((got["depth_map"].cpu() - depth).abs().max()) < 2e-4
    mse = float(((got["rgb_map"].cpu() - rgb) ** 2).mean())
    assert mse < 1e-9          # PSNR > 90 dB (north-star bar: 45 dB)
