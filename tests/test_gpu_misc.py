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
