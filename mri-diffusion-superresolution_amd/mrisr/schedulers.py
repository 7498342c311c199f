"""Host-side scheduler tables: the three members the reference touches - ``alphas_cumprod``
(``src/adapters/res_srdiff.py:13,60``), ``set_timesteps(n, device=)`` (``:53``), ``timesteps`` (``:54``) - with the
diffusers DDPM/DDIM table conventions (config keys nb ResDif c11:44-46).  Tiny, host-only; the
per-step arithmetic runs in the fused HIP step kernels driven by ``mrisr.pipeline``.

Every diffusers option that changes the table or the step is either implemented or refused: nothing is swallowed.
The reference's training config sets ``prediction_type="epsilon"``, ``timestep_spacing="trailing"`` and
``rescale_betas_zero_snr=True`` (nb ResDif c11:44-46)."""
from __future__ import annotations

import numpy as np
import torch

# options whose diffusers DEFAULT is what the fused step kernels implement; any other value raises
_FIXED = {
    "variance_type": ("fixed_small",),
    "clip_sample": (False,),          # x0 clipping is a Sampler argument (clip_sample_range), not scheduler state
    "thresholding": (False,),
    "set_alpha_to_one": (False,),     # the last DDIM step uses alphas_cumprod[0]
    "trained_betas": (None,),
    "dynamic_thresholding_ratio": (0.995,),
    "sample_max_value": (1.0,),
    "clip_sample_range": (1.0,),
}


def rescale_zero_terminal_snr(betas: torch.Tensor) -> torch.Tensor:
    """Lin et al. 2023 ("Common diffusion noise schedules and sample steps are flawed"), Algorithm 1 - what diffusers applies
    for ``rescale_betas_zero_snr=True``: shift sqrt(abar) so that the last entry is exactly 0, rescale so that the first is
    unchanged, and turn the result back into betas."""
    abar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    first, last = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
    abar_sqrt = (abar_sqrt - last) * (first / (first - last))
    abar = abar_sqrt ** 2
    alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
    return 1.0 - alphas


class DDPMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", timestep_spacing: str = "leading", steps_offset: int = 0,
                 prediction_type: str = "epsilon", rescale_betas_zero_snr: bool = False, **options):
        if prediction_type != "epsilon":
            raise ValueError("only epsilon prediction is used by the reference (nb ResDif c11:44)")
        for k, v in options.items():
            if k not in _FIXED:
                raise ValueError(f"unknown scheduler option {k!r}")
            if v not in _FIXED[k]:
                raise ValueError(f"scheduler option {k}={v!r} is not supported (the fused step implements {k}={_FIXED[k][0]!r})")
        self.num_train_timesteps = num_train_timesteps
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise ValueError(f"unknown beta_schedule {beta_schedule}")
        if rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        self.rescale_betas_zero_snr = bool(rescale_betas_zero_snr)
        self.betas = betas
        # with a zero terminal SNR the last entry is exactly 0: the forward shift (res_srdiff.py:13-25) is fine with that, the
        # reverse step divides by sqrt(abar_t) (:86) - the C sampler clamps abar_t to 2^-24 there
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        if timestep_spacing not in ("leading", "trailing"):
            raise ValueError(f"unknown timestep_spacing {timestep_spacing}")
        self.timestep_spacing = timestep_spacing
        self.steps_offset = steps_offset
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    def set_timesteps(self, num_inference_steps: int, device=None):
        T, n = self.num_train_timesteps, num_inference_steps
        if n > T:
            raise ValueError("num_inference_steps > num_train_timesteps")
        if self.timestep_spacing == "leading":
            ts = (np.arange(0, n) * (T // n)).round()[::-1].astype(np.int64) + self.steps_offset
        else:  # "trailing"
            ts = np.round(np.arange(T, 0, -T / n)).astype(np.int64) - 1
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy(ts.copy())
        if device is not None:
            self.timesteps = self.timesteps.to(device)


class DDIMScheduler(DDPMScheduler):
    """eta = 0, set_alpha_to_one=False, no clipping - the sampler BASELINE.json names."""
