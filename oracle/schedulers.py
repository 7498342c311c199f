"""Oracle: scheduler tables and steps (numpy, fp64 tables -> fp32 like diffusers).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Spec: SURVEY.md App. A.7 (diffusers DDPMScheduler/DDIMScheduler; un-vendored).  Scheduler surface
the reference touches: ``.alphas_cumprod`` (res_srdiff.py:13,60), ``.set_timesteps(n, device=)``
(:53), ``.timesteps`` (:54).  Config keys: nb ResDif c11:44-46.
"""
from __future__ import annotations

import numpy as np
import torch


def make_betas(num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
               schedule: str = "scaled_linear") -> np.ndarray:
    if schedule == "scaled_linear":  # SD-1.5
        # diffusers builds the table in float32
        b = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    elif schedule == "linear":  # MNIST notebook c5:1-9 uses linear 1e-4 -> 0.02
        b = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    else:
        raise ValueError(schedule)
    return b.numpy()


def rescale_zero_terminal_snr(betas: np.ndarray) -> np.ndarray:
    """``rescale_betas_zero_snr=True`` (nb ResDif c11:46): Algorithm 1 of Lin et al. 2023, as diffusers applies it to the
    float32 beta table - sqrt(abar) shifted to end at exactly 0 and rescaled to keep its first entry."""
    b = torch.from_numpy(betas)
    s = torch.cumprod(1.0 - b, dim=0).sqrt()
    s0, sT = s[0].clone(), s[-1].clone()
    s = (s - sT) * (s0 / (s0 - sT))
    abar = s ** 2
    alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
    return (1.0 - alphas).numpy()


def alphas_cumprod_from_betas(betas: np.ndarray) -> np.ndarray:
    return torch.cumprod(1.0 - torch.from_numpy(betas), dim=0).numpy()


def make_timesteps(n: int, num_train_timesteps: int = 1000, spacing: str = "leading",
                   steps_offset: int = 0) -> np.ndarray:
    """``leading``: arange(n)*(T//n) reversed + offset;  ``trailing``: round(arange(T,0,-T/n)) - 1."""
    T = num_train_timesteps
    if spacing == "leading":
        ts = (np.arange(0, n) * (T // n)).round()[::-1].astype(np.int64) + steps_offset
    elif spacing == "trailing":
        ts = np.round(np.arange(T, 0, -T / n)).astype(np.int64) - 1
    else:
        raise ValueError(spacing)
    return ts.copy()


class OracleScheduler:
    """The three members the reference uses, plus ddim_step for BASELINE's 50-step DDIM."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", timestep_spacing="leading", steps_offset=0, rescale_betas_zero_snr=False):
        self.num_train_timesteps = num_train_timesteps
        self.betas = make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        if rescale_betas_zero_snr:
            self.betas = rescale_zero_terminal_snr(self.betas)
        self.alphas_cumprod = torch.from_numpy(alphas_cumprod_from_betas(self.betas))
        self.timestep_spacing, self.steps_offset = timestep_spacing, steps_offset
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self.num_inference_steps = None

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy(make_timesteps(n, self.num_train_timesteps, self.timestep_spacing,
                                                         self.steps_offset))
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def ddim_coeffs(self, t: int):
        """x_prev = c_x * x_t + c_e * eps  (eta=0, set_alpha_to_one=False, no clipping)."""
        ac = self.alphas_cumprod.double()
        t_prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = ac[t]
        a_p = ac[t_prev] if t_prev >= 0 else ac[0]
        c_x = (a_p / a_t).sqrt()
        c_e = (1 - a_p).sqrt() - (a_p * (1 - a_t) / a_t).sqrt()
        return float(c_x), float(c_e)

    def ddim_step(self, eps: torch.Tensor, t: int, x: torch.Tensor) -> torch.Tensor:
        ac = self.alphas_cumprod.to(x.dtype)
        t_prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = ac[t]
        a_p = ac[t_prev] if t_prev >= 0 else ac[0]
        x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        return a_p.sqrt() * x0 + (1 - a_p).sqrt() * eps

    def ddpm_step(self, eps: torch.Tensor, t: int, x: torch.Tensor, noise=None, clip_sample_range: float = 0.0) -> torch.Tensor:
        """Ancestral step of diffusers' DDPMScheduler.step (un-vendored; variance_type "fixed_small", epsilon prediction;
        BASELINE config 1 "10-step DDPM").  ``noise`` is the z ~ N(0, I) the scheduler would draw (None: mean only)."""
        ac = self.alphas_cumprod.to(x.dtype)
        t_prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = ac[t]
        a_p = ac[t_prev] if t_prev >= 0 else torch.ones((), dtype=x.dtype)
        alpha_t = a_t / a_p
        beta_t = 1 - alpha_t
        x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        if clip_sample_range > 0:
            x0 = x0.clamp(-clip_sample_range, clip_sample_range)
        mean = (a_p.sqrt() * beta_t / (1 - a_t)) * x0 + (alpha_t.sqrt() * (1 - a_p) / (1 - a_t)) * x
        if t > 0 and noise is not None:
            var = ((1 - a_p) / (1 - a_t) * beta_t).clamp(min=1e-20)
            mean = mean + var.sqrt() * noise
        return mean
