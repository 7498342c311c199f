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
