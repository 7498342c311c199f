"""GPU: whole-model parity of the HIP UNet / ControlNet / adapter / sampler against the CPU oracle and the committed
golden vectors.  f32 path: 1e-3 relative (north_star tolerance, written here); bf16 path: relative-L2 bound."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def maxrel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.fixture(scope="module")
def tiny():
    from oracle import unet as ou
    cfg = ou.TINY
    up = ou.init_unet_params(cfg, seed=101, perturb_norm=True)
    lora = ou.init_lora_params(up, rank=4, seed=103)
    cp = ou.init_controlnet_params(cfg, seed=102, perturb_norm=True)
    return cfg, up, lora, cp


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("bf16", 5e-2)])
@pytest.mark.parametrize("lora_fused", [True, False])
def test_unet_forward_matches_oracle(tiny, dt, tol, lora_fused):
    import mrisr
    from oracle import unet as ou
    cfg, up, lora, _ = tiny
    p = {**up, **lora}
    g = torch.Generator().manual_seed(5)
    B, h = 2, 16
    x = torch.randn((B, 4, h, h), generator=g)
    ctx = torch.randn((B, 77, cfg.cross_attention_dim), generator=g)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype=dt, lora_rank=4, lora_alpha=4, lora_fused=lora_fused,
                                     flash_attention=True)
    net.load_state_dict(p)
    assert net.num_parameters == ou.count_params(p)
    for t in (torch.tensor(801), torch.tensor([10, 990])):
        ref = ou.unet_forward(p, cfg, x, t, ctx)
        out = net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
        assert out.shape == ref.shape
        assert rel(out, ref) < tol, (dt, rel(out, ref))
        if dt == "f32":
            assert maxrel(out, ref) < 1e-3
    # return_dict=False -> tuple, and cached context (ehs=None) reproduces the same output
    out2 = net(x.cuda(), torch.tensor(801).cuda(), encoder_hidden_states=None, return_dict=False)[0]
    ref = ou.unet_forward(p, cfg, x, torch.tensor(801), ctx)
    assert rel(out2, ref) < tol


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("bf16", 5e-2)])
def test_unet_with_every_gemm_split(tiny, dt, tol):
    """Split-K forced on every GEMM that can be split: the reduce kernel then runs every epilogue variant (bias, time
    embedding, residual in place, LoRA rank tail, head-major Q/K scatter and the transposed V^T store - the latter once
    used a lane exchange that is only valid in the MFMA kernels' lane layout)."""
    import ctypes as C
    import mrisr
    from mrisr import _lib as L
    from oracle import unet as ou
    cfg, up, lora, _ = tiny
    p = {**up, **lora}
    g = torch.Generator().manual_seed(15)
    x = torch.randn((2, 4, 16, 16), generator=g)
    ctx = torch.randn((2, 77, cfg.cross_attention_dim), generator=g)
    t = torch.tensor([3, 600])
    ref = ou.unet_forward(p, cfg, x, t, ctx)
    lib = L.lib()
    try:
        for split in (2, 4):
            lib.mrisr_debug_force_split(C.c_int(split))
            net = mrisr.UNet2DConditionModel(cfg, compute_dtype=dt, lora_rank=4, lora_alpha=4)
            net.load_state_dict(p)
            out = net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
            assert rel(out, ref) < tol, (dt, split, rel(out, ref))
    finally:
        lib.mrisr_debug_force_split(C.c_int(0))


def test_staged_head_scatter_and_mfma_lora_match_the_direct_paths():
    """bf16 GEMM epilogues have two forms each: head-major Q / K / V^T outputs staged through LDS vs stored straight from the
    accumulator layout (flag 16), and the rank-4 LoRA up-projection on the matrix cores vs in the scalar epilogue (flag 8).
    Staging must not change a single bit; the matrix-core form rounds z and s*B to bf16 first, so both forms are measured against
    the f32 engine and must be equally close.  Head sizes 40 / 80 (SD-1.5 levels 0 / 1) so that every alignment condition of the staged path holds."""
    import ctypes as C
    import mrisr
    from mrisr import _lib as L
    from mrisr import params as P
    cfg = mrisr.UNetConfig(block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                           layers_per_block=1)
    dev = torch.device("cuda")
    sd = P.random_state_dict(P.unet_param_shapes(cfg), 77, dev)
    sd.update(P.random_state_dict(P.lora_param_shapes(cfg, 4), 78, dev))
    g = torch.Generator(device=dev).manual_seed(79)
    x = torch.randn((4, 4, 32, 32), generator=g, device=dev)
    ctx = torch.randn((4, 77, cfg.cross_attention_dim), generator=g, device=dev)
    t = torch.tensor([1, 250, 600, 999], device=dev)
    lib = L.lib()
    outs = {}
    try:
        for flags in (0, 16, 8):
            lib.mrisr_debug_gemm_flags(C.c_int(flags))
            net = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4)
            net.load_state_dict(sd)
            outs[flags] = net(x, t, encoder_hidden_states=ctx).sample.float().clone()
            torch.cuda.synchronize()
    finally:
        lib.mrisr_debug_gemm_flags(C.c_int(0))
    assert torch.isfinite(outs[0]).all() and float(outs[0].abs().mean()) > 1e-3
    assert torch.equal(outs[0], outs[16])
    ref = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4)
    ref.load_state_dict(sd)
    f32 = ref(x, t, encoder_hidden_states=ctx).sample.float()
    e_mma, e_scalar = rel(outs[0], f32), rel(outs[8], f32)
    print(f"bf16 vs f32: matrix-core LoRA {e_mma:.4e}, scalar-epilogue LoRA {e_scalar:.4e}, between them {rel(outs[0], outs[8]):.4e}")
    assert e_mma < 5e-2 and e_scalar < 5e-2
    assert e_mma < 1.25 * e_scalar + 1e-3, (e_mma, e_scalar)   # rounding z / s*B to bf16 does not cost accuracy against f32


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("bf16", 5e-2)])
def test_unet_materialised_attention_path(tiny, dt, tol):
    import mrisr
    from oracle import unet as ou
    cfg, up, _, _ = tiny
    g = torch.Generator().manual_seed(6)
    x = torch.randn((1, 4, 8, 8), generator=g)
    ctx = torch.randn((1, 77, cfg.cross_attention_dim), generator=g)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype=dt, flash_attention=False)
    net.load_state_dict(up)
    ref = ou.unet_forward(up, cfg, x, torch.tensor(3), ctx)
    assert rel(net(x.cuda(), 3, encoder_hidden_states=ctx.cuda()).sample, ref) < tol


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("bf16", 5e-2)])
def test_controlnet_and_residual_injection(tiny, dt, tol):
    import mrisr
    from oracle import unet as ou
    cfg, up, lora, cp = tiny
    p = {**up, **lora}
    g = torch.Generator().manual_seed(7)
    B, h = 2, 8
    x = torch.randn((B, 4, h, h), generator=g)
    ctx = torch.randn((B, 77, cfg.cross_attention_dim), generator=g)
    cond = torch.randn((B, 3, 8 * h, 8 * h), generator=g)
    t = torch.tensor(500)
    dref, mref = ou.controlnet_forward(cp, cfg, x, t, ctx, cond)
    cnet = mrisr.ControlNetModel(cfg, compute_dtype=dt)
    cnet.load_state_dict(cp)
    down, mid = cnet(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda(), controlnet_cond=cond.cuda(), return_dict=False)
    assert len(down) == 12 and [tuple(d.shape) for d in down] == [tuple(d.shape) for d in dref]
    for a, b in zip(list(down) + [mid], dref + [mref]):
        assert rel(a, b) < tol
    # feed the ORACLE's residuals into the HIP UNet: isolates the injection path (incl. the last-skip/mid aliasing)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype=dt, lora_rank=4, lora_alpha=4)
    net.load_state_dict(p)
    ref = ou.unet_forward(p, cfg, x, t, ctx, dref, mref)
    out = net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda(), down_block_additional_residuals=[d.cuda() for d in dref],
              mid_block_additional_residual=mref.cuda()).sample
    assert rel(out, ref) < tol
    with pytest.raises(ValueError):
        cnet(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda(), controlnet_cond=cond[:, :, :32, :32].cuda())


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("bf16", 3e-2)])
def test_adapter_matches_reference_golden_and_feeds_unet(tiny, golden_dir, dt, tol):
    import mrisr
    from oracle import adapter as oad
    from oracle import unet as ou
    g = np.load(os.path.join(golden_dir, "adapter_xl.npz"))
    acfg = oad.ADAPTER_TINY
    ap = oad.init_adapter_params(acfg, seed=401)
    ad = mrisr.Adapter_XL(channels=acfg.channels, nums_rb=acfg.nums_rb, cin=acfg.cin, ksize=acfg.ksize, sk=True,
                          use_conv=True, compute_dtype=dt)
    ad.load_state_dict(ap)
    feats = ad(torch.from_numpy(g["x"]).cuda())
    for i, f in enumerate(feats):  # golden = the reference's own Adapter_XL output
        assert rel(f, torch.from_numpy(g[f"feat{i}"])) < tol
    # features into the UNet (diffusers down_intrablock_additional_residuals convention)
    cfg, up, _, _ = tiny
    gen = torch.Generator().manual_seed(9)
    x = torch.randn((2, 4, 8, 8), generator=gen)
    ctx = torch.randn((2, 77, cfg.cross_attention_dim), generator=gen)
    fo = oad.adapter_forward(ap, acfg, torch.from_numpy(g["x"]))
    ref = ou.unet_forward(up, cfg, x, torch.tensor(77), ctx, down_intrablock_additional_residuals=fo)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype=dt)
    net.load_state_dict(up)
    out = net(x.cuda(), 77, encoder_hidden_states=ctx.cuda(), down_intrablock_additional_residuals=[f.cuda() for f in fo]).sample
    assert rel(out, ref) < (1e-3 if dt == "f32" else 5e-2)
    # the LAST skip carries feature 3 (diffusers' in-place `sample +=` on res_samples[-1]): with a strong feature 3 the
    # product follows the in-place oracle and is far from the out-of-place reading (skip 11 without the feature)
    big = [f.clone() for f in fo]
    big[3] = big[3] * 32.0
    ref_in = ou.unet_forward(up, cfg, x, torch.tensor(77), ctx, down_intrablock_additional_residuals=[f.clone() for f in big])
    ref_out = ou.unet_forward(up, cfg, x, torch.tensor(77), ctx, down_intrablock_additional_residuals=[f.clone() for f in big],
                              _adapter_last_skip_inplace=False)
    out = net(x.cuda(), 77, encoder_hidden_states=ctx.cuda(), down_intrablock_additional_residuals=[f.cuda() for f in big]).sample
    tl = 1e-3 if dt == "f32" else 5e-2
    assert rel(out, ref_in) < tl
    if dt == "f32":  # the two readings differ by 2.3e-2 here: distinguishable at the f32 tolerance, not at the bf16 one
        assert rel(ref_out, ref_in) > 1e-2 and rel(out, ref_out) > 1e-2


def test_forward_shift_matches_reference_golden(golden_dir):
    import mrisr
    g = np.load(os.path.join(golden_dir, "res_shift_forward.npz"))
    sched = mrisr.DDPMScheduler()
    hr, lr, nz = (torch.from_numpy(g[k]).cuda() for k in ("hr", "lr", "noise"))
    out_s = mrisr.get_res_shifting_latents(hr, lr, torch.from_numpy(g["t_scalar"]), sched, nz)
    out_b = mrisr.get_res_shifting_latents(hr, lr, torch.from_numpy(g["t_batch"]), sched, nz)
    np.testing.assert_allclose(out_s.cpu().numpy(), g["out_scalar"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out_b.cpu().numpy(), g["out_batch"], rtol=1e-5, atol=1e-6)


class _StubVAE:
    class config:
        scaling_factor = 0.18215

    def encode(self, x):
        z = torch.nn.functional.avg_pool2d(x[:, :1], 8).repeat(1, 4, 1, 1)
        return type("E", (), {"latent_dist": type("D", (), {"sample": staticmethod(lambda: z)})})

    def decode(self, z):
        return type("O", (), {"sample": torch.nn.functional.interpolate(z.mean(1, keepdim=True), scale_factor=8.0, mode="nearest")})


@pytest.mark.parametrize("tag", ["n5", "n20"])
def test_log_validation_matches_reference_trajectory(tiny, golden_dir, tag):
    """mrisr.log_validation (HIP ControlNet+UNet, fused Res-SRDiff step, hipGraph) reproduces the reference's own
    log_validation run (golden made by importing /root/reference) - f32 parity mode."""
    import mrisr
    g = np.load(os.path.join(golden_dir, f"log_validation_{tag}.npz"))
    cfg, up, lora, cp = tiny
    n = int(g["n_steps"])
    unet = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4)
    unet.load_state_dict({**up, **lora})
    cnet = mrisr.ControlNetModel(cfg, compute_dtype="f32")
    cnet.load_state_dict(cp)
    gen = torch.Generator().manual_seed(201)
    base = torch.randn((1, 1, 32, 32), generator=gen)
    hr = torch.nn.functional.interpolate(base, size=(512, 512), mode="bicubic", align_corners=False).clamp(-1, 1)
    lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 4), scale_factor=4.0, mode="bilinear")
    ctx = torch.randn((1, 77, cfg.cross_attention_dim), generator=torch.Generator().manual_seed(301)).cuda()
    sched = mrisr.DDPMScheduler(timestep_spacing="leading", steps_offset=1)
    acc = type("A", (), {"device": torch.device("cuda")})
    # same global-RNG draws, in the same order, as the reference (CPU generator -> identical values)
    torch.manual_seed(int(g["seed"]))
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: torch.randn(t.shape).to(t.device, t.dtype)  # draw on the CPU stream
    try:
        panel = np.asarray(mrisr.log_validation(unet, cnet, _StubVAE(), [{"hr": hr, "lr": lr}], sched, torch.float32, acc,
                                                ctx, num_inference_steps=n))
    finally:
        torch.randn_like = orig
    assert list(panel.shape) == g["panel_shape"].tolist()
    W = panel.shape[1] // 3
    small = panel[:, W:2 * W][::8, ::8, 0]
    assert np.abs(small.astype(int) - g["gen_panel_small"].astype(int)).max() <= 1


def test_reference_eager_loop_over_product_models_matches_reference_trajectory(tiny, golden_dir):
    """the reference's OWN per-step loop (res_srdiff.py:63-96; here its pinned restatement
    oracle.sampler.res_srdiff_sample, which is checked against the same golden on the CPU) keeps working when it is handed the
    product's duck-typed models: a 0-dim device `t`, `return_dict=False` -> (tuple of 12 residuals, mid), `.sample`, torch
    arithmetic on the device between the calls.  Every state of the reference's N=5 run."""
    import mrisr
    from oracle import sampler as osa
    g = np.load(os.path.join(golden_dir, "log_validation_n5.npz"))
    cfg, up, lora, cp = tiny
    unet = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4)
    unet.load_state_dict({**up, **lora})
    cnet = mrisr.ControlNetModel(cfg, compute_dtype="f32")
    cnet.load_state_dict(cp)
    gen = torch.Generator().manual_seed(201)
    base = torch.randn((1, 1, 32, 32), generator=gen)
    hr = torch.nn.functional.interpolate(base, size=(512, 512), mode="bicubic", align_corners=False).clamp(-1, 1)
    lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 4), scale_factor=4.0, mode="bilinear")
    ctx = torch.randn((1, 77, cfg.cross_attention_dim), generator=torch.Generator().manual_seed(301)).cuda()
    lr_lat = (torch.nn.functional.avg_pool2d(lr, 8).repeat(1, 4, 1, 1) * 0.18215).cuda()
    cond = mrisr.prepare_condition_image(lr.cuda())
    torch.manual_seed(int(g["seed"]))
    init_noise = torch.randn(lr_lat.shape).cuda()
    step_noise = [torch.randn(lr_lat.shape).cuda() for _ in range(4)]
    sched = mrisr.DDPMScheduler(timestep_spacing="leading", steps_offset=1)
    sched.set_timesteps(5, device="cuda")
    seen = []

    def unet_spy(x, t, **kw):  # what the loop hands over: 0-dim int64 device t, a tuple of 12 device residuals + mid
        seen.append((t.ndim, t.device.type, t.dtype, len(kw["down_block_additional_residuals"]),
                     tuple(kw["mid_block_additional_residual"].shape)))
        return unet(x, t, **kw)

    traj = osa.res_srdiff_sample(unet_spy, cnet, lr_lat, ctx[0:1], cond, sched.timesteps, sched.alphas_cumprod, init_noise, step_noise)
    assert len(traj) == 6 and all(s == (0, "cuda", torch.int64, 12, (1, cfg.block_out_channels[-1], 8, 8)) for s in seen), seen  # 512^2 px -> 64^2 latents -> 8^2 mid
    assert g["states"].shape[0] == 5  # the golden holds the state BEFORE each of the 5 steps; the final one is scored through the panel
    for k, x in enumerate(traj[:5]):
        assert x.is_cuda and x.dtype == torch.float32
        r = rel(x, torch.from_numpy(g["states"][k]))
        assert r < 1e-3, (k, r)  # north_star: 1e-3 rel f32
    gen = mrisr.decode_to_vis(traj[5], _StubVAE())  # res_srdiff.py:100 on the loop's result
    assert np.abs(gen[::8, ::8, 0].astype(int) - g["gen_panel_small"].astype(int)).max() <= 1


def test_sampler_trajectory_states_and_graph_equals_eager(tiny, golden_dir):
    import mrisr
    g = np.load(os.path.join(golden_dir, "log_validation_n5.npz"))
    cfg, up, lora, cp = tiny
    unet = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4)
    unet.load_state_dict({**up, **lora})
    cnet = mrisr.ControlNetModel(cfg, compute_dtype="f32")
    cnet.load_state_dict(cp)
    gen = torch.Generator().manual_seed(201)
    base = torch.randn((1, 1, 32, 32), generator=gen)
    hr = torch.nn.functional.interpolate(base, size=(512, 512), mode="bicubic", align_corners=False).clamp(-1, 1)
    lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 4), scale_factor=4.0, mode="bilinear")
    ctx = torch.randn((1, 77, cfg.cross_attention_dim), generator=torch.Generator().manual_seed(301)).cuda()
    lr_lat = (torch.nn.functional.avg_pool2d(lr, 8).repeat(1, 4, 1, 1) * 0.18215).cuda()
    cond = mrisr.prepare_condition_image(lr.cuda())
    torch.manual_seed(int(g["seed"]))
    init_noise = torch.randn(lr_lat.shape)
    step_noise = torch.stack([torch.randn(lr_lat.shape) for _ in range(4)])
    # run k steps at a time with truncated schedules and compare the state before step k with the golden
    full = mrisr.DDPMScheduler(timestep_spacing="leading", steps_offset=1)
    full.set_timesteps(5)
    x0 = mrisr.get_res_shifting_latents(lr_lat, lr_lat, full.timesteps[0], full, init_noise.cuda())
    np.testing.assert_allclose(x0.cpu().numpy(), g["states"][0], rtol=1e-5, atol=1e-6)
    finals = {}
    for use_graph in (True, False):
        lat = x0.clone().contiguous()
        mrisr.Sampler(unet, full, cnet, kind="resshift").run(lat, ctx, lr_latents=lr_lat, step_noise=step_noise.cuda(),
                                                             controlnet_cond=cond, use_graph=use_graph)
        torch.cuda.synchronize()
        finals[use_graph] = lat.cpu()
    assert torch.equal(finals[True], finals[False])  # graph replay == eager launches, bit for bit
    # every intermediate state of the reference run: stop the fused loop after k steps (mrisr_sampler_set_range)
    for k in range(1, 5):
        lat = x0.clone().contiguous()
        smp = mrisr.Sampler(unet, full, cnet, kind="resshift")
        smp.set_range(0, k)
        smp.run(lat, ctx, lr_latents=lr_lat, step_noise=step_noise.cuda(), controlnet_cond=cond)
        torch.cuda.synchronize()
        ref = torch.from_numpy(g["states"][k])
        assert rel(lat, ref) < 1e-3 and maxrel(lat, ref) < 1e-3, (k, rel(lat, ref))


def test_config1_mnist_plumbing_ddim10(golden_dir):
    """BASELINE config 1 ("MNIST_Super_Resolution.ipynb path"): 1-channel 2-level UNet, 28x28 padded to 32x32, bs=8,
    10 steps, linear betas 1e-4..0.02 (nb MNIST c5:1-9).  HIP sampler (f32) vs the CPU oracle loop."""
    import mrisr
    from oracle import sampler as osa
    from oracle import schedulers as osch
    from oracle import unet as ou
    cfg = ou.MNIST
    p = ou.init_unet_params(cfg, seed=55, perturb_norm=True)
    g = torch.Generator().manual_seed(56)
    digits = torch.randn((8, 1, 28, 28), generator=g)
    x = torch.nn.functional.pad(digits, (2, 2, 2, 2))  # 28 -> 32 (divisible by 2^(levels-1))
    ctx = torch.randn((8, 77, cfg.cross_attention_dim), generator=g)
    so = osch.OracleScheduler(beta_start=1e-4, beta_end=0.02, beta_schedule="linear")
    so.set_timesteps(10)
    traj = osa.ddim_sample(ou.OracleUNet(p, cfg), x, ctx, so)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32")
    net.load_state_dict(p)
    sp = mrisr.DDIMScheduler(beta_start=1e-4, beta_end=0.02, beta_schedule="linear")
    sp.set_timesteps(10)
    lat = x.cuda().clone().contiguous()
    mrisr.Sampler(net, sp, kind="ddim").run(lat, ctx.cuda())
    torch.cuda.synchronize()
    assert rel(lat, traj[-1]) < 1e-3 and maxrel(lat, traj[-1]) < 1e-3
    assert lat[:, :, 2:30, 2:30].shape == (8, 1, 28, 28)


@pytest.mark.parametrize("clip", [0.0, 1.0])
def test_config1_mnist_plumbing_ddpm10(clip):
    """BASELINE config 1 as worded ("10-step DDPM, bs=8"): the ancestral step with per-step noise, with and without diffusers'
    default x0 clipping, on the 1-channel 2-level UNet.  HIP sampler (f32, graph replay) vs the CPU oracle loop, 1e-3 rel."""
    import mrisr
    from oracle import sampler as osa
    from oracle import schedulers as osch
    from oracle import unet as ou
    cfg = ou.MNIST
    p = ou.init_unet_params(cfg, seed=57, perturb_norm=True)
    g = torch.Generator().manual_seed(58)
    x = torch.randn((8, 1, 32, 32), generator=g)
    ctx = torch.randn((8, 77, cfg.cross_attention_dim), generator=g)
    z = torch.randn((10, 8, 1, 32, 32), generator=g)
    so = osch.OracleScheduler(beta_start=1e-4, beta_end=0.02, beta_schedule="linear")
    so.set_timesteps(10)
    traj = osa.ddpm_sample(ou.OracleUNet(p, cfg), x, ctx, so, z, clip)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32")
    net.load_state_dict(p)
    sp = mrisr.DDPMScheduler(beta_start=1e-4, beta_end=0.02, beta_schedule="linear")
    sp.set_timesteps(10)
    for graph in (True, False):
        lat = x.cuda().clone().contiguous()
        mrisr.Sampler(net, sp, kind="ddpm", clip_sample_range=clip).run(lat, ctx.cuda(), step_noise=z.cuda(), use_graph=graph)
        torch.cuda.synchronize()
        assert rel(lat, traj[-1]) < 1e-3 and maxrel(lat, traj[-1]) < 1e-3, (graph, rel(lat, traj[-1]))
    # the noise matters (checked on the device side: a second oracle loop would double this test's CPU time)
    lat0 = x.cuda().clone().contiguous()
    mrisr.Sampler(net, sp, kind="ddpm", clip_sample_range=clip).run(lat0, ctx.cuda(), step_noise=torch.zeros_like(z).cuda(), use_graph=False)
    assert not torch.allclose(lat0, lat)


def test_sampler_kind_errors():
    import mrisr
    from oracle import unet as ou
    net = mrisr.UNet2DConditionModel(ou.MNIST, compute_dtype="f32")
    net.load_state_dict(ou.init_unet_params(ou.MNIST, seed=1))
    sp = mrisr.DDIMScheduler()
    sp.set_timesteps(4)
    with pytest.raises(ValueError):
        mrisr.Sampler(net, sp, kind="euler")
    with pytest.raises(RuntimeError):  # x0 clipping is a DDPM option
        L = mrisr._lib
        L.check(L.lib().mrisr_sampler_set_clip(mrisr.Sampler(net, sp, kind="ddim")._h, 1.0))


@pytest.mark.parametrize("split", [2, 4])
def test_groupnorm_sums_the_split_k_slabs_itself_and_changes_no_bit(tiny, split):
    """conv1 -> GroupNorm2 of every resnet: when K is split for the conv, the GroupNorm kernel sums the f32 slabs itself (the reduce
    kernel's operations in its order: bias, time-embedding row, bf16 rounding) instead of a reduce launch followed by a GroupNorm launch.
    Same bits as the two launches (mrisr_debug_gn_slabs(0)), per-sample timesteps and a scalar one; the launch counts say which ran."""
    import ctypes as C
    import json
    import mrisr
    from mrisr import _lib as L
    from oracle import unet as ou
    cfg, up, lora, _ = tiny
    p = {**up, **lora}
    g = torch.Generator().manual_seed(16)
    x = torch.randn((2, 4, 16, 16), generator=g)
    ctx = torch.randn((2, 77, cfg.cross_attention_dim), generator=g)
    lib = L.lib()

    def run(net, t):
        net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda())
        lib.mrisr_prof_reset(); lib.mrisr_prof_enable(1)
        out = net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample.clone()
        torch.cuda.synchronize(); lib.mrisr_prof_enable(0)
        buf = C.create_string_buffer(1 << 20)
        n = lib.mrisr_prof_report(buf, len(buf))
        cls = json.loads(buf.value[:n].decode())
        lib.mrisr_prof_reset()
        return out, cls
    try:
        lib.mrisr_debug_force_split(C.c_int(split))
        net = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4)
        net.load_state_dict(p)
        for t in (torch.tensor([3, 600]), torch.tensor(77)):
            lib.mrisr_debug_gn_slabs(C.c_int(0))
            two, c0 = run(net, t)
            lib.mrisr_debug_gn_slabs(C.c_int(1))
            one, c1 = run(net, t)
            n_fused = c1.get("groupnorm_from_slabs", {}).get("launches", 0)
            assert "groupnorm_from_slabs" not in c0 and n_fused >= 8, (n_fused, sorted(c1))
            assert c0["splitk_reduce"]["launches"] - c1["splitk_reduce"]["launches"] == n_fused
            assert torch.equal(one, two), float((one.float() - two.float()).abs().max())
        ref = ou.unet_forward(p, cfg, x, torch.tensor([3, 600]), ctx)
        lib.mrisr_debug_gn_slabs(C.c_int(1))
        assert rel(net(x.cuda(), torch.tensor([3, 600]).cuda(), encoder_hidden_states=ctx.cuda()).sample, ref) < 5e-2
    finally:
        lib.mrisr_debug_force_split(C.c_int(0))
        lib.mrisr_debug_gn_slabs(C.c_int(-1))
