"""GPU: AutoencoderKL encode / decode of libmrisr against the CPU oracle (f32: 1e-3 relative; bf16: relative-L2 bound), and
the host mirror's diffusers-like surface (latent_dist.sample / mode, decode(...).sample, scaling_factor)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


@pytest.fixture(scope="module")
def tiny():
    from oracle import vae as ov
    return ov.TINY_VAE, ov.init_vae_params(ov.TINY_VAE, seed=77)


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("bf16", 4e-2)])
@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 64, 128)])
def test_vae_encode_decode_match_oracle(tiny, dt, tol, B, H, W):
    import mrisr
    from oracle import vae as ov
    cfg, p = tiny
    g = torch.Generator().manual_seed(9)
    x = torch.randn((B, 3, H, W), generator=g).clamp(-1, 1)
    vae = mrisr.AutoencoderKL(cfg, compute_dtype=dt)
    vae.load_state_dict(p)
    assert vae.num_parameters == ov.count_params(p)
    mom_ref = ov.encode_moments(p, cfg, x)
    dist = vae.encode(x.cuda()).latent_dist
    assert dist.parameters.shape == mom_ref.shape
    assert rel(dist.parameters, mom_ref) < tol, rel(dist.parameters, mom_ref)
    assert torch.equal(dist.mode(), dist.mean)
    # sampling consumes the generator like diffusers' DiagonalGaussianDistribution.sample
    gs = torch.Generator(device="cuda").manual_seed(5)
    z = dist.sample(generator=gs)
    gs2 = torch.Generator(device="cuda").manual_seed(5)
    noise = torch.randn(dist.mean.shape, generator=gs2, device="cuda")
    assert rel(z, ov.sample_latents(dist.parameters.float().cpu(), noise.cpu())) < 1e-5
    zin = torch.randn((B, 4, H // 8, W // 8), generator=g)
    img_ref = ov.decode(p, cfg, zin)
    img = vae.decode(zin.cuda()).sample
    assert img.shape == img_ref.shape and rel(img, img_ref) < tol, rel(img, img_ref)
    assert vae.decode(zin.cuda(), return_dict=False)[0].shape == img_ref.shape


def test_vae_reference_call_pattern_and_errors(tiny):
    """The two reference lines (res_srdiff.py:50,110): encode(...).latent_dist.sample() * scaling_factor, and
    decode(latents / scaling_factor).sample; plus the shape errors."""
    import mrisr
    from oracle import vae as ov
    cfg, p = tiny
    vae = mrisr.AutoencoderKL(cfg, compute_dtype="f32").eval()
    with pytest.raises(mrisr.MrisrError, match="load_state_dict"):
        vae.encode(torch.zeros(1, 3, 64, 64).cuda())
    vae.load_state_dict(p)
    assert abs(vae.config.scaling_factor - 0.18215) < 1e-9
    x = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(1)) * 2 - 1
    lat = vae.encode(x.cuda()).latent_dist.sample() * vae.config.scaling_factor
    img = vae.decode(lat / vae.config.scaling_factor).sample
    assert img.shape == (1, 3, 64, 64) and torch.isfinite(img).all()
    with pytest.raises(ValueError, match="image must be"):
        vae.encode(torch.zeros(1, 1, 64, 64).cuda())
    with pytest.raises(ValueError, match="divisible"):
        vae.encode(torch.zeros(1, 3, 60, 64).cuda())
    bad = dict(p)
    del bad["decoder.conv_out.bias"]
    v2 = mrisr.AutoencoderKL(cfg, compute_dtype="f32")
    v2.load_state_dict(bad)  # bias is optional in the packer; a missing WEIGHT is an error:
    del bad["decoder.conv_out.weight"]
    v3 = mrisr.AutoencoderKL(cfg, compute_dtype="f32")
    with pytest.raises(mrisr.MrisrError, match="missing parameter: decoder.conv_out.weight"):
        v3.load_state_dict(bad)


class _OracleVAE:
    """The oracle VAE behind the same duck-typed surface (CPU): swapped in for mrisr.AutoencoderKL to check integration."""

    def __init__(self, cfg, p):
        self.cfg, self.p = cfg, p
        self.config = type("C", (), {"scaling_factor": cfg.scaling_factor})

    def encode(self, x):
        from oracle import vae as ov
        mom = ov.encode_moments(self.p, self.cfg, x.float().cpu()).to(x.device)
        mean, logvar = mom.chunk(2, 1)

        class D:
            def sample(_s):
                return mean + torch.exp(0.5 * logvar.clamp(-30, 20)) * torch.randn(mean.shape).to(mean.device)
        return type("E", (), {"latent_dist": D()})

    def decode(self, z):
        from oracle import vae as ov
        return type("O", (), {"sample": ov.decode(self.p, self.cfg, z.float().cpu()).to(z.device)})


def test_log_validation_with_device_vae(tiny):
    """The reference's validation sampler end to end on the device (VAE encode -> ControlNet + UNet loop -> VAE decode ->
    uint8 panel, res_srdiff.py:35-116): mrisr.AutoencoderKL in place of the VAE changes the panel by at most 1 LSB
    against the CPU oracle VAE behind the same surface."""
    import numpy as np
    import mrisr
    from oracle import unet as ou
    cfg, p = tiny
    ucfg = ou.TINY
    up = ou.init_unet_params(ucfg, seed=101, perturb_norm=True)
    cp = ou.init_controlnet_params(ucfg, seed=102, perturb_norm=True)
    unet = mrisr.UNet2DConditionModel(ucfg, compute_dtype="f32")
    unet.load_state_dict(up)
    cnet = mrisr.ControlNetModel(ucfg, compute_dtype="f32")
    cnet.load_state_dict(cp)
    vae = mrisr.AutoencoderKL(cfg, compute_dtype="f32")
    vae.load_state_dict(p)
    gen = torch.Generator().manual_seed(11)
    hr = torch.nn.functional.interpolate(torch.randn((1, 1, 32, 32), generator=gen), size=(512, 512), mode="bicubic").clamp(-1, 1)
    lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 4), scale_factor=4.0, mode="bilinear")
    ctx = torch.randn((1, 77, ucfg.cross_attention_dim), generator=gen).cuda()
    sched = mrisr.DDPMScheduler(timestep_spacing="leading", steps_offset=1)
    acc = type("A", (), {"device": torch.device("cuda")})
    panels = []
    orig = (torch.randn_like, torch.randn)
    for v in (vae, _OracleVAE(cfg, p)):
        torch.manual_seed(123)
        torch.randn_like = lambda t, **kw: orig[1](t.shape).to(t.device, t.dtype)  # every draw from the CPU stream
        try:
            if v is vae:  # the device posterior draws through torch.randn(device=cuda): route it to the CPU stream too
                torch.randn = lambda *a, **kw: orig[1](*a, **{k: w for k, w in kw.items() if k not in ("device", "generator")}).to(kw.get("device", "cpu"))
            panels.append(np.asarray(mrisr.log_validation(unet, cnet, v, [{"hr": hr, "lr": lr}], sched, torch.float32, acc, ctx,
                                                          num_inference_steps=4)).astype(int))
        finally:
            torch.randn_like, torch.randn = orig
    assert panels[0].shape == panels[1].shape
    assert np.abs(panels[0] - panels[1]).max() <= 1
