"""Host-side scheduler tables: the three members the reference touches - ``alphas_cumprod``
(``src/adapters/res_srdiff.py:13,60``), ``set_timesteps(n, device=)`` (``:53``), ``timesteps`` (``:54``) - with the
diffusers DDPM/DDIM table conventions (config keys nb ResDif c11:44-46).  Tiny, host-only; the
per-step arithmetic runs in the fused HIP step kernels driven by ``mrisr.pipeline``."""
from __future__ import annotations

import numpy as np
import torch


class DDPMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", timestep_spacing: str = "leading", steps_offset: int = 0,
                 prediction_type: str = "epsilon", **_ignored):
        if prediction_type != "epsilon":
            raise ValueError("only epsilon prediction is used by the reference (nb ResDif c11:44)")
        self.num_train_timesteps = num_train_timesteps
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise ValueError(f"unknown beta_schedule {beta_schedule}")
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
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
        elif self.timestep_spacing == "trailing":
            ts = np.round(np.arange(T, 0, -T / n)).astype(np.int64) - 1
        else:
            raise ValueError(f"unknown timestep_spacing {self.timestep_spacing}")
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy(ts.copy())
        if device is not None:
            self.timesteps = self.timesteps.to(device)


class DDIMScheduler(DDPMScheduler):
    """eta = 0, set_alpha_to_one=False, no clipping - the sampler BASELINE.json names."""
