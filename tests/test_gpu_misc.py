"""GPU: sampler step kernels vs the oracle formulas (incl. trailing spacing with t = 999), boundary dtypes, error
behaviour of the host mirror."""
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_resshift_steps_match_oracle_formulas():
    """Fused Res-SRDiff step kernel (device coefficient table + device step counter) vs oracle.sampler
    (= reference res_srdiff.py:84-96), trailing spacing, with and without the stochastic term."""
    import mrisr
    from oracle import sampler as osa
    from oracle import schedulers as osch
    from oracle import unet as ou
    cfg = ou.TINY
    p = ou.init_unet_params(cfg, seed=77)
    g = torch.Generator().manual_seed(78)
    lr = torch.randn((2, 4, 8, 8), generator=g)
    ctx = torch.randn((2, 77, cfg.cross_attention_dim), generator=g)
    noise = torch.randn((3, 2, 4, 8, 8), generator=g)
    init = torch.randn((2, 4, 8, 8), generator=g)
    so = osch.OracleScheduler(timestep_spacing="trailing")
    so.set_timesteps(4)
    sp = mrisr.DDPMScheduler(timestep_spacing="trailing")
    sp.set_timesteps(4)
    assert torch.equal(so.timesteps, sp.timesteps) and int(sp.timesteps[0]) == 999
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32")
    net.load_state_dict(p)
    x_T = osa.res_shift_forward(lr, lr, so.timesteps[0], so.alphas_cumprod, init)
    for nz in (noise, None):
        steps = [noise[i] for i in range(3)] if nz is not None else [torch.zeros_like(lr)] * 3
        traj = osa.res_srdiff_sample(ou.OracleUNet(p, cfg), None, lr, ctx, None, so.timesteps.tolist(),
                                     so.alphas_cumprod, init, steps)
        lat = x_T.cuda().contiguous()
        mrisr.Sampler(net, sp, kind="resshift").run(lat, ctx.cuda(), lr_latents=lr.cuda(),
                                                    step_noise=nz.cuda() if nz is not None else None)
        torch.cuda.synchronize()
        assert float((lat.cpu() - traj[-1]).norm() / traj[-1].norm()) < 1e-3


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_boundary_dtypes(dt):
    """sample / output tensors may arrive in any of the reference's weight dtypes (fp32, fp16 autocast, bf16)."""
    import mrisr
    from oracle import unet as ou
    cfg = ou.TINY
    p = ou.init_unet_params(cfg, seed=79)
    g = torch.Generator().manual_seed(80)
    x = torch.randn((1, 4, 8, 8), generator=g)
    ctx = torch.randn((1, 77, cfg.cross_attention_dim), generator=g)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32")
    net.load_state_dict(p)
    ref = ou.unet_forward(p, cfg, x.to(dt).float(), torch.tensor(5), ctx.to(dt).float())
    out = net(x.to(dt).cuda(), 5, encoder_hidden_states=ctx.to(dt).cuda()).sample
    assert out.dtype == dt
    tol = 1e-3 if dt == torch.float32 else 2e-2
    assert float((out.float().cpu() - ref).norm() / ref.norm()) < tol


def test_error_behaviour():
    import mrisr
    from oracle import unet as ou
    cfg = ou.TINY
    p = ou.init_unet_params(cfg, seed=81)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16")
    with pytest.raises(mrisr.MrisrError):  # forward before load_state_dict
        net(torch.zeros(1, 4, 8, 8).cuda(), 1, encoder_hidden_states=torch.zeros(1, 77, 64).cuda())
    bad = dict(p)
    bad.pop("mid_block.resnets.0.conv1.weight")
    with pytest.raises(mrisr.MrisrError, match="missing parameter"):
        net.load_state_dict(bad)
    net.load_state_dict(p)
    with pytest.raises(ValueError):  # wrong channel count
        net(torch.zeros(1, 3, 8, 8).cuda(), 1, encoder_hidden_states=torch.zeros(1, 77, 64).cuda())
    with pytest.raises(mrisr.MrisrError):  # latent size not divisible by 2^(levels-1)
        net(torch.zeros(1, 4, 12, 12).cuda(), 1, encoder_hidden_states=torch.zeros(1, 77, 64).cuda())
    fresh = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16")
    fresh.load_state_dict(p)
    with pytest.raises(mrisr.MrisrError):  # no context given and none cached
        fresh(torch.zeros(1, 4, 8, 8).cuda(), 1)


def test_condition_image_and_vis_match_reference_golden(golden_dir):
    """mrisr.prepare_condition_image / decode_to_vis (the product's own glue, on the device) against the vectors the reference's
    res_srdiff.py:27-33 / :107-122 produced (tests/golden/make_golden.py)."""
    import os

    import numpy as np

    import mrisr
    g = np.load(os.path.join(golden_dir, "condition_and_vis.npz"))
    cond = mrisr.prepare_condition_image(torch.from_numpy(g["img"]).cuda(), target_size=(64, 64))
    assert cond.shape == (2, 3, 64, 64) and cond.is_cuda
    np.testing.assert_allclose(cond.cpu().numpy(), g["cond"], rtol=1e-5, atol=1e-6)
    same = mrisr.prepare_condition_image(torch.zeros(1, 3, 16, 16).cuda(), target_size=(16, 16))
    assert list(same.shape) == g["cond3_shape"].tolist()
    assert mrisr.prepare_condition_image(torch.zeros(1, 1, 8, 8).cuda()).shape == (1, 3, 512, 512)  # the reference's default size
    vis = mrisr.decode_to_vis(torch.from_numpy(g["dec"]).cuda(), None, is_latent=False)
    assert vis.dtype == np.uint8 and vis.shape == g["vis"].shape
    assert np.abs(vis.astype(int) - g["vis"].astype(int)).max() == 0


def test_long_lived_sampler_survives_workspace_replans():
    """a Sampler's captured graph bakes in the model's workspace addresses.  A forward at another batch, or a
    training step, re-plans (and may reallocate) that workspace; the next run() of the SAME sampler must re-capture instead of
    replaying a graph that points into freed memory.  Result == the eager loop, bit for bit, every time."""
    import mrisr
    from oracle import unet as ou
    cfg = ou.TINY
    p = ou.init_unet_params(cfg, seed=91, perturb_norm=True)
    p.update(ou.init_lora_params(p, rank=4, seed=92))
    g = torch.Generator().manual_seed(93)
    x1 = torch.randn((1, 4, 16, 16), generator=g)
    c1 = torch.randn((1, 77, cfg.cross_attention_dim), generator=g)
    x4 = torch.randn((4, 4, 32, 32), generator=g)
    c4 = torch.randn((4, 77, cfg.cross_attention_dim), generator=g)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
    net.load_state_dict(p)
    sp = mrisr.DDIMScheduler(timestep_spacing="leading", steps_offset=1)
    sp.set_timesteps(3)
    A = mrisr.Sampler(net, sp, kind="ddim")
    lat_buf = torch.empty((1, 4, 16, 16), device="cuda")

    def run(sampler, graph):
        lat_buf.copy_(x1)
        sampler.run(lat_buf, c1.cuda(), use_graph=graph)
        torch.cuda.synchronize()
        return lat_buf.cpu().clone()

    want = run(mrisr.Sampler(net, sp, kind="ddim"), False)
    assert torch.equal(run(A, True), want)
    net(x4.cuda(), torch.tensor([5, 50, 500, 950]).cuda(), encoder_hidden_states=c4.cuda())   # bigger geometry: re-plan + realloc
    torch.cuda.synchronize()
    assert torch.equal(run(A, True), want)
    B2 = mrisr.Sampler(net, sp, kind="ddim")                                               # a second sampler at a larger batch
    lat4 = x4.cuda().clone().contiguous()
    B2.run(lat4, c4.cuda(), use_graph=True)
    torch.cuda.synchronize()
    assert torch.equal(run(A, True), want)
    tr = mrisr.LoRATrainer(net, lr=0.0)                                                     # a training step: keep=1 arena growth
    tr.forward_backward(x4.cuda(), torch.tensor([5, 50, 500, 950]), c4.cuda(), torch.zeros_like(x4).cuda())
    torch.cuda.synchronize()
    assert torch.equal(run(A, True), want)


def test_sampler_operand_shapes_are_checked():
    """the step kernels index lr[i] and noise[step * n + i] unchecked, so the C ABI refuses operands that do not cover
    the latents; get_res_shifting_latents refuses timesteps the reference's indexing would refuse."""
    import mrisr
    from oracle import unet as ou
    cfg = ou.TINY
    p = ou.init_unet_params(cfg, seed=94)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32")
    net.load_state_dict(p)
    cnet = mrisr.ControlNetModel(cfg, compute_dtype="f32")
    cnet.load_state_dict(ou.init_controlnet_params(cfg, seed=95))
    ctx = torch.zeros(2, 77, cfg.cross_attention_dim).cuda()
    lat = torch.zeros(2, 4, 8, 8).cuda()
    sp = mrisr.DDPMScheduler(timestep_spacing="leading", steps_offset=1)
    sp.set_timesteps(4)   # resshift: steps 0..2 are stochastic -> 3 slabs
    rs = mrisr.Sampler(net, sp, kind="resshift")
    with pytest.raises(mrisr.MrisrError, match="fewer slabs"):
        rs.run(lat, ctx, lr_latents=lat.clone(), step_noise=torch.zeros(2, 2, 4, 8, 8).cuda())
    rs.run(lat, ctx, lr_latents=lat.clone(), step_noise=torch.zeros(3, 2, 4, 8, 8).cuda())
    rs.set_range(0, 2)    # only steps 0-1 run: two slabs are enough
    rs.run(lat, ctx, lr_latents=lat.clone(), step_noise=torch.zeros(2, 2, 4, 8, 8).cuda())
    rs.set_range(0, 4)
    with pytest.raises(mrisr.MrisrError, match="lr_latents"):
        rs.run(lat, ctx, lr_latents=torch.zeros(1, 4, 8, 8).cuda(), step_noise=None)
    with pytest.raises(mrisr.MrisrError, match="step_noise"):
        rs.run(lat, ctx, lr_latents=lat.clone(), step_noise=torch.zeros(3, 2, 4, 8, 7).cuda())
    # diffusers DDPM with steps_offset=1: t = 1 > 0 on the LAST step too, so all n_steps slabs are read
    dp = mrisr.Sampler(net, sp, kind="ddpm")
    with pytest.raises(mrisr.MrisrError, match="fewer slabs"):
        dp.run(lat, ctx, step_noise=torch.zeros(3, 2, 4, 8, 8).cuda())
    dp.run(lat, ctx, step_noise=torch.zeros(4, 2, 4, 8, 8).cuda())
    with pytest.raises(mrisr.MrisrError, match="encoder_hidden_states"):
        L = mrisr._lib
        import ctypes as C
        t_lat, t_e = L.as_tensor(lat), L.as_tensor(torch.zeros(3, 77, cfg.cross_attention_dim).cuda())
        L.check(L.lib().mrisr_sampler_run(dp._h, C.byref(t_lat), None, None, C.byref(t_e), None, None, 0, 0, L.stream_ptr()))
    cs = mrisr.Sampler(net, sp, cnet, kind="ddim")
    with pytest.raises(mrisr.MrisrError, match="controlnet_cond"):
        cs.run(lat, ctx, controlnet_cond=torch.zeros(1, 3, 64, 64).cuda())   # batch 1 vs latents batch 2
    with pytest.raises(mrisr.MrisrError, match="controlnet_cond"):
        cs.run(lat, ctx, controlnet_cond=torch.zeros(2, 3, 32, 32).cuda())   # not 8h x 8w
    torch.cuda.synchronize()
    # forward shift: timestep length / range, as the reference's alphas_cumprod[timesteps] would raise
    hr = torch.zeros(3, 4, 8, 8).cuda()
    with pytest.raises(RuntimeError):
        mrisr.get_res_shifting_latents(hr, hr, torch.tensor([1, 2]), sp)
    with pytest.raises(IndexError):
        mrisr.get_res_shifting_latents(hr, hr, torch.tensor(1000), sp)
    with pytest.raises(IndexError):
        mrisr.get_res_shifting_latents(hr, hr, torch.tensor([0, 5, 1000]), sp)
    a = mrisr.get_res_shifting_latents(hr + 1, hr, torch.tensor(-1), sp, torch.zeros_like(hr))     # torch semantics: last entry
    b = mrisr.get_res_shifting_latents(hr + 1, hr, torch.tensor(999), sp, torch.zeros_like(hr))
    assert torch.equal(a, b)


def test_reference_training_schedule_trailing_zero_snr():
    """The reference's own scheduler config (nb ResDif c11:44-46: trailing spacing + rescale_betas_zero_snr) samples t = 999
    first, where abar = 0 exactly and res_srdiff.py:86 divides by sqrt(abar).  The C sampler clamps abar_t to 2^-24 (SURVEY.md
    App. C.4); with the same clamp applied to the oracle's table the two loops agree, and the result is finite."""
    import mrisr
    from oracle import sampler as osa
    from oracle import schedulers as osch
    from oracle import unet as ou
    cfg = ou.TINY
    p = ou.init_unet_params(cfg, seed=96, perturb_norm=True)
    g = torch.Generator().manual_seed(97)
    lr = 0.2 * torch.randn((2, 4, 8, 8), generator=g)
    ctx = torch.randn((2, 77, cfg.cross_attention_dim), generator=g)
    noise = torch.randn((3, 2, 4, 8, 8), generator=g)
    init = torch.randn((2, 4, 8, 8), generator=g)
    so = osch.OracleScheduler(timestep_spacing="trailing", rescale_betas_zero_snr=True)
    so.set_timesteps(4)
    sp = mrisr.DDPMScheduler(timestep_spacing="trailing", rescale_betas_zero_snr=True, prediction_type="epsilon")
    sp.set_timesteps(4)
    assert int(sp.timesteps[0]) == 999 and float(sp.alphas_cumprod[999]) == 0.0
    x_T = osa.res_shift_forward(lr, lr, so.timesteps[0], so.alphas_cumprod, init)   # abar = 0: x_T = LR + noise, no division
    got = mrisr.get_res_shifting_latents(lr.cuda(), lr.cuda(), sp.timesteps[0], sp, init.cuda())
    assert torch.allclose(got.cpu(), x_T, rtol=1e-5, atol=1e-6)
    clamped = so.alphas_cumprod.clamp_min(2.0 ** -24)
    traj = osa.res_srdiff_sample(ou.OracleUNet(p, cfg), None, lr, ctx, None, so.timesteps.tolist(), clamped, init,
                                 [noise[i] for i in range(3)])
    # (the oracle's first state is built from the clamped table too; start both loops from the same x_T instead)
    x = x_T
    for i, t in enumerate(so.timesteps.tolist()):
        eps = ou.unet_forward(p, cfg, x, torch.tensor(t), ctx)
        tp = int(so.timesteps[i + 1]) if i + 1 < 4 else 0
        x = osa.res_shift_reverse_step(x, eps, lr, clamped[t], clamped[tp], noise[i] if tp > 0 else None)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32")
    net.load_state_dict(p)
    lat = x_T.cuda().contiguous()
    mrisr.Sampler(net, sp, kind="resshift").run(lat, ctx.cuda(), lr_latents=lr.cuda(), step_noise=noise.cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(lat).all()
    assert float((lat.cpu() - x).norm() / x.norm()) < 1e-3
    assert len(traj) == 5


@pytest.mark.parametrize("dt,tol", [("f32", 2e-6), ("bf16", 5e-3)])
def test_sampler_time_embedding_table_matches_the_per_step_embedding(dt, tol):
    """The fused sampler computes the time embedding of every step of a run once (sinusoid -> MLP -> the per-resnet projections for all rows
    at once, f32 inputs as in the one-row kernel) and copies one row per step; against computing it inside every step (mrisr_debug_temb_table(0)):
    f32 identical; bf16: the two GEMV kernels sum in a different order, and a 1e-7 change of an embedding flips bf16 roundings downstream - the
    5-step latents of this small model then differ by ~2e-3, the same as for any other f32-level perturbation of the bf16 engine.  Graph and eager agree bit for bit either
    way, a sub-range of the schedule indexes the table from its own first row, and a plain forward after the run computes its own embedding."""
    import ctypes as C
    import mrisr
    from mrisr import _lib as L
    from oracle import unet as ou
    cfg = ou.TINY
    p = ou.init_unet_params(cfg, seed=191, perturb_norm=True)
    g = torch.Generator().manual_seed(193)
    x = torch.randn((2, 4, 16, 16), generator=g)
    ctx = torch.randn((2, 77, cfg.cross_attention_dim), generator=g)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype=dt)
    net.load_state_dict(p)
    sp = mrisr.DDIMScheduler(timestep_spacing="leading", steps_offset=1)
    sp.set_timesteps(5)
    lib = L.lib()

    def run(graph):
        lat = x.cuda().clone().contiguous()
        mrisr.Sampler(net, sp, kind="ddim").run(lat, ctx.cuda(), use_graph=graph)
        torch.cuda.synchronize()
        return lat.cpu()
    try:
        lib.mrisr_debug_temb_table(C.c_int(0))
        per_step = run(True)
        before = net(x.cuda(), torch.tensor(321).cuda(), encoder_hidden_states=ctx.cuda()).sample.clone()
        lib.mrisr_debug_temb_table(C.c_int(1))
        tab_graph, tab_eager = run(True), run(False)
        after = net(x.cuda(), torch.tensor(321).cuda(), encoder_hidden_states=ctx.cuda()).sample.clone()
    finally:
        lib.mrisr_debug_temb_table(C.c_int(-1))
    assert torch.equal(tab_graph, tab_eager)
    assert torch.equal(before, after), "a plain forward after a sampler run must not see the sampler's table"
    d = float((tab_graph - per_step).norm() / per_step.norm())
    print(f"time-embedding table vs per step [{dt}]: {d:.3e}")
    assert d < tol, d
