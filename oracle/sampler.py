"""Oracle: the reference's Res-SRDiff shift / reverse step / validation sampler, plus the DDIM loop
BASELINE.json names.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates ``/root/reference/src/adapters/res_srdiff.py`` (whole file):
  forward shift  :7-25    condition image :27-33    sampler loop :35-105    uint8 panel :107-122
Pinned by tests/golden/res_srdiff_*.npz, which were produced by importing the reference itself
(tests/golden/make_golden.py).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def res_shift_forward(hr: torch.Tensor, lr: torch.Tensor, t, alphas_cumprod: torch.Tensor,
                      noise: torch.Tensor) -> torch.Tensor:
    """x_t = sqrt(a_t) HR + (1 - sqrt(a_t)) LR + sqrt(1 - a_t) eps      (res_srdiff.py:13-25).
    ``t`` may be a 0-dim or a [B] int64 tensor."""
    a = alphas_cumprod.to(hr.device)[t].reshape(-1, 1, 1, 1)
    ra = a ** 0.5
    return ra * hr + (1 - ra) * lr + (1 - a) ** 0.5 * noise


def condition_image(img: torch.Tensor, target_size=(512, 512)) -> torch.Tensor:
    """1 -> 3 channel expand + bilinear(align_corners=False) resize   (res_srdiff.py:27-33)."""
    if img.shape[1] == 1:
        img = img.expand(-1, 3, -1, -1)
    if tuple(img.shape[-2:]) != tuple(target_size):
        img = F.interpolate(img, size=target_size, mode="bilinear", align_corners=False)
    return img


def res_shift_reverse_step(x_t, eps, lr, a_t, a_prev, noise: Optional[torch.Tensor]):
    """res_srdiff.py:84-96.  ``noise`` is None on the last step (prev_t == 0)."""
    x0 = (x_t - (1 - a_t ** 0.5) * lr - (1 - a_t) ** 0.5 * eps) / (a_t ** 0.5)
    x = (a_prev ** 0.5) * x0 + (1 - a_prev ** 0.5) * lr
    if noise is not None:
        x = x + ((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)) ** 0.5 * noise
    return x


def to_uint8_panel(decoded: torch.Tensor) -> np.ndarray:
    """(x/2+.5).clamp(0,1) -> HWC uint8 of sample 0, grey -> 3 channels   (res_srdiff.py:113-121)."""
    img = (decoded / 2 + 0.5).clamp(0, 1).cpu().permute(0, 2, 3, 1).numpy()
    u8 = (img[0] * 255).astype(np.uint8)
    if u8.shape[-1] == 1:
        u8 = np.concatenate([u8] * 3, axis=-1)
    return u8


def res_srdiff_sample(unet: Callable, controlnet: Optional[Callable], lr_latents: torch.Tensor,
                      ctx: torch.Tensor, control_image: Optional[torch.Tensor], timesteps: Sequence[int],
                      alphas_cumprod: torch.Tensor, init_noise: torch.Tensor,
                      step_noise: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """The loop of res_srdiff.py:58-96 with all noise supplied by the caller.
    Returns the state BEFORE every step plus the final state (len(timesteps)+1 tensors)."""
    ts = [int(t) for t in timesteps]
    dev = lr_latents.device  # the reference keeps timesteps / alphas_cumprod on accelerator.device (:53-60): 0-dim device t
    alphas_cumprod = alphas_cumprod.to(dev)
    x = res_shift_forward(lr_latents, lr_latents, torch.tensor(ts[0], device=dev), alphas_cumprod, init_noise)
    traj = [x]
    k = 0
    for i, t in enumerate(ts):
        tt = torch.tensor(t, dtype=torch.int64, device=dev)
        down = mid = None
        if controlnet is not None:
            down, mid = controlnet(x, tt, encoder_hidden_states=ctx, controlnet_cond=control_image,
                                   return_dict=False)
        eps = unet(x, tt, encoder_hidden_states=ctx, down_block_additional_residuals=down,
                   mid_block_additional_residual=mid).sample
        prev_t = ts[i + 1] if i + 1 < len(ts) else 0
        a_t = alphas_cumprod[t].to(x.dtype)
        a_p = alphas_cumprod[prev_t].to(x.dtype)
        noise = None
        if prev_t > 0:
            noise = step_noise[k]
            k += 1
        x = res_shift_reverse_step(x, eps, lr_latents, a_t, a_p, noise)
        traj.append(x)
    return traj


def ddim_sample(unet: Callable, x_T: torch.Tensor, ctx: torch.Tensor, scheduler,
                controlnet: Optional[Callable] = None, control_image: Optional[torch.Tensor] = None,
                intrablock: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
    """BASELINE metric loop: n-step DDIM(eta=0) (absent from the reference)."""
    x = x_T
    traj = [x]
    for t in scheduler.timesteps.tolist():
        tt = torch.tensor(t, dtype=torch.int64)
        down = mid = None
        if controlnet is not None:
            down, mid = controlnet(x, tt, encoder_hidden_states=ctx, controlnet_cond=control_image,
                                   return_dict=False)
        eps = unet(x, tt, encoder_hidden_states=ctx, down_block_additional_residuals=down,
                   mid_block_additional_residual=mid,
                   down_intrablock_additional_residuals=(list(intrablock) if intrablock is not None else None)
                   ).sample
        x = scheduler.ddim_step(eps, t, x)
        traj.append(x)
    return traj


def ddpm_sample(unet: Callable, x_T: torch.Tensor, ctx: torch.Tensor, scheduler, step_noise: Optional[torch.Tensor] = None,
                clip_sample_range: float = 0.0) -> List[torch.Tensor]:
    """BASELINE config 1 loop: n-step ancestral DDPM with the linear-beta table of nb MNIST c5:1-9; ``step_noise[i]`` is the
    draw of step i (unused on the t == 0 step)."""
    x = x_T
    traj = [x]
    for i, t in enumerate(scheduler.timesteps.tolist()):
        eps = unet(x, torch.tensor(t, dtype=torch.int64), encoder_hidden_states=ctx).sample
        x = scheduler.ddpm_step(eps, t, x, step_noise[i] if step_noise is not None else None, clip_sample_range)
        traj.append(x)
    return traj
