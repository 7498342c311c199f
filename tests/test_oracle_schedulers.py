"""CPU checks of the oracle's scheduler steps (no GPU): closed-form properties of the ancestral DDPM step and of DDIM that
hold for any table, so the restatement of the un-vendored diffusers arithmetic is pinned by identities
rather than by its own output."""
import torch

from oracle import schedulers as osch


def _sched(n):
    s = osch.OracleScheduler(beta_start=1e-4, beta_end=0.02, beta_schedule="linear")  # nb MNIST c5:1-9
    s.set_timesteps(n)
    return s


def test_linear_beta_table_matches_the_notebook_definition():
    s = _sched(10)
    b = torch.linspace(1e-4, 0.02, 1000)
    assert torch.equal(torch.from_numpy(s.betas), b)
    assert torch.allclose(s.alphas_cumprod, torch.cumprod(1 - b, 0))
    assert s.timesteps.tolist() == [900, 800, 700, 600, 500, 400, 300, 200, 100, 0]


def test_ddpm_step_is_the_gaussian_posterior_mean():
    """With the true noise as the prediction, the step returns q(x_{t-1} | x_t, x_0)'s mean; on the last step (t_prev < 0,
    abar_prev = 1) that is x_0 itself."""
    s = _sched(10)
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn((2, 1, 8, 8), generator=g, dtype=torch.float64)
    e = torch.randn((2, 1, 8, 8), generator=g, dtype=torch.float64)
    ac = s.alphas_cumprod.double()
    for t in s.timesteps.tolist():
        xt = ac[t].sqrt() * x0 + (1 - ac[t]).sqrt() * e
        out = s.ddpm_step(e, t, xt)
        tp = t - 100
        a_p = ac[tp] if tp >= 0 else torch.ones((), dtype=torch.float64)
        al = ac[t] / a_p
        want = (a_p.sqrt() * (1 - al) * x0 + al.sqrt() * (1 - a_p) * xt) / (1 - ac[t])
        assert torch.allclose(out, want, rtol=1e-10, atol=1e-12)
        if tp < 0:
            assert torch.allclose(out, x0, rtol=1e-9, atol=1e-9)


def test_ddpm_step_variance_and_clip():
    s = _sched(10)
    x = torch.full((1, 1, 2, 2), 3.0, dtype=torch.float64)
    e = torch.zeros_like(x)
    z = torch.ones_like(x)
    ac = s.alphas_cumprod.double()
    t = 500
    var = (1 - ac[400]) / (1 - ac[500]) * (1 - ac[500] / ac[400])
    assert torch.allclose(s.ddpm_step(e, t, x, z) - s.ddpm_step(e, t, x), var.sqrt() * z)
    assert torch.equal(s.ddpm_step(e, 0, x, z), s.ddpm_step(e, 0, x))  # no noise on the t == 0 step
    clipped = s.ddpm_step(e, t, x, None, 1.0)
    x0c = (x / ac[t].sqrt()).clamp(-1, 1)
    al = ac[500] / ac[400]
    assert torch.allclose(clipped, (ac[400].sqrt() * (1 - al) * x0c + al.sqrt() * (1 - ac[400]) * x) / (1 - ac[500]))


def test_ddim_step_inverts_the_forward_process():
    """eta = 0 DDIM with the true noise moves x_t to the same (x_0, eps) pair at t_prev."""
    s = osch.OracleScheduler()
    s.set_timesteps(50)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn((2, 4, 4, 4), generator=g, dtype=torch.float64)
    e = torch.randn((2, 4, 4, 4), generator=g, dtype=torch.float64)
    ac = s.alphas_cumprod.double()
    for t in (980, 500, 20):
        xt = ac[t].sqrt() * x0 + (1 - ac[t]).sqrt() * e
        tp = t - 20
        want = ac[tp].sqrt() * x0 + (1 - ac[tp]).sqrt() * e
        assert torch.allclose(s.ddim_step(e, t, xt), want, rtol=1e-9, atol=1e-9)
