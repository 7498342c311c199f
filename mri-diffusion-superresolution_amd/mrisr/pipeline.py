"""Host-side mirror of the reference's sampler glue (``src/adapters/res_srdiff.py``): same function names, argument
meaning and error behaviour; the arithmetic runs in libmrisr.so.

  get_res_shifting_latents  :7-25      prepare_condition_image :27-33
  log_validation            :35-105    decode_to_vis           :107-122

plus ``sample`` - the timestep loop itself (``:63-96``) as ONE call into the C-ABI sampler, which captures a step
(ControlNet -> UNet -> fused reverse step) into a hipGraph and replays it; no host sync on ``prev_t > 0``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib as L
from .models import ControlNetModel, UNet2DConditionModel


def get_res_shifting_latents(hr_latents, lr_latents, timesteps, scheduler, noise=None):
    """x_t = sqrt(a_t) HR + (1 - sqrt(a_t)) LR + sqrt(1 - a_t) eps   (reference res_srdiff.py:7-25)."""
    dev = hr_latents.device
    if noise is None:
        noise = torch.randn_like(hr_latents)
    ac = scheduler.alphas_cumprod.to(device=dev, dtype=torch.float32).contiguous()
    t = torch.as_tensor(timesteps).to(device=dev, dtype=torch.int64).contiguous()
    # the reference indexes ``alphas_cumprod[timesteps]`` and broadcasts [B,1,1,1] (:13-19): a wrong length or an index outside
    # the table is an error there; the kernel reads ac[t[b]] unchecked, so both are checked here
    if t.ndim > 1 or (t.ndim == 1 and t.numel() not in (1, hr_latents.shape[0])):
        raise RuntimeError(f"timesteps of shape {tuple(t.shape)} do not broadcast against a batch of {hr_latents.shape[0]}")
    if t.numel() and (int(t.min()) < -ac.numel() or int(t.max()) >= ac.numel()):
        raise IndexError(f"timestep out of range for an alphas_cumprod table of {ac.numel()} entries")
    if t.numel() and int(t.min()) < 0:
        t = torch.where(t < 0, t + ac.numel(), t)  # torch indexing semantics for negative indices
    hr, lr, nz = (x.to(torch.float32).contiguous() for x in (hr_latents, lr_latents, noise))
    out = torch.empty_like(hr)
    t_hr, t_lr, t_nz, t_t, t_out = (L.as_tensor(x) for x in (hr, lr, nz, t, out))
    L.check(L.lib().mrisr_resshift_forward(C.byref(t_hr), C.byref(t_lr), C.byref(t_nz), C.c_void_p(ac.data_ptr()),
                                           C.byref(t_t), C.byref(t_out), L.stream_ptr()))
    return out.to(hr_latents.dtype)


def prepare_condition_image(image, target_size=(512, 512)):
    """1 -> 3 channel expand + bilinear resize (reference res_srdiff.py:27-33).  Once per slice, outside the loop:
    tensor plumbing, done with torch on the device."""
    if image.shape[1] == 1:
        image = image.expand(-1, 3, -1, -1)
    if tuple(image.shape[-2:]) != tuple(target_size):
        image = F.interpolate(image, size=target_size, mode="bilinear", align_corners=False)
    return image


def decode_to_vis(data, vae, is_latent=True):
    """reference res_srdiff.py:107-122."""
    decoded = vae.decode(data / vae.config.scaling_factor).sample if is_latent else data
    img = (decoded / 2 + 0.5).clamp(0, 1).cpu().permute(0, 2, 3, 1).float().numpy()
    img_np = (img[0] * 255).astype(np.uint8)
    if img_np.shape[-1] == 1:
        img_np = np.concatenate([img_np] * 3, axis=-1)
    return img_np


class Sampler:
    """Owns a C-ABI sampler (device tables + the captured step graph) for one (unet, controlnet, schedule)."""

    def __init__(self, unet: UNet2DConditionModel, scheduler, controlnet: Optional[ControlNetModel] = None,
                 kind: str = "ddim", clip_sample_range: float = 0.0):
        """``kind``: "ddim" (eta 0), "resshift" (res_srdiff.py:84-96) or "ddpm" (ancestral, diffusers DDPMScheduler.step with
        "fixed_small" variance; ``clip_sample_range`` > 0 clips the predicted x0 as diffusers' ``clip_sample`` does)."""
        kinds = {"ddim": L.STEP_DDIM, "resshift": L.STEP_RESSHIFT, "ddpm": L.STEP_DDPM}
        if kind not in kinds:
            raise ValueError(f"unknown sampler kind {kind!r}")
        self.unet, self.controlnet, self.kind = unet, controlnet, kind
        ts = scheduler.timesteps.detach().cpu().to(torch.int64).numpy().copy()
        ac = scheduler.alphas_cumprod.detach().cpu().to(torch.float32).numpy().copy()
        self.n_steps = len(ts)
        self._h = C.c_void_p()
        L.check(L.lib().mrisr_sampler_create(unet._h, controlnet._h if controlnet is not None else None,
                                             kinds[kind],
                                             ts.ctypes.data_as(C.c_void_p), int(len(ts)),
                                             ac.ctypes.data_as(C.c_void_p), int(len(ac)), C.byref(self._h)))
        if clip_sample_range > 0:
            L.check(L.lib().mrisr_sampler_set_clip(self._h, float(clip_sample_range)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                L.lib().mrisr_sampler_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def set_range(self, first_step: int, last_step: int):
        """Run only steps [first_step, last_step) of the schedule on the next ``run`` (resume / inspection)."""
        L.check(L.lib().mrisr_sampler_set_range(self._h, int(first_step), int(last_step)))

    def run(self, latents: torch.Tensor, encoder_hidden_states: torch.Tensor, lr_latents: Optional[torch.Tensor] = None,
            step_noise: Optional[torch.Tensor] = None, controlnet_cond: Optional[torch.Tensor] = None,
            adapter_features: Optional[Sequence[torch.Tensor]] = None, use_graph: bool = True) -> torch.Tensor:
        """Advance ``latents`` [B,C,h,w] (f32, updated IN PLACE) through every timestep."""
        if latents.dtype != torch.float32 or not latents.is_contiguous():
            raise ValueError("latents must be contiguous float32 (updated in place)")
        dev = latents.device
        ehs = encoder_hidden_states.to(dev).contiguous()
        if ehs.shape[0] != latents.shape[0]:
            ehs = ehs.expand(latents.shape[0], -1, -1).contiguous()
        lr = lr_latents.to(dev, torch.float32).contiguous() if lr_latents is not None else None
        nz = step_noise.to(dev, torch.float32).contiguous() if step_noise is not None else None
        cond = controlnet_cond.to(dev).contiguous() if controlnet_cond is not None else None
        feats = [f.to(dev).contiguous() for f in (adapter_features or [])]
        keep = (ehs, lr, nz, cond, feats)  # noqa: F841  keep alive until the stream has consumed them
        t_lat, t_e = L.as_tensor(latents), L.as_tensor(ehs)
        t_lr = L.as_tensor(lr) if lr is not None else None
        # step noise is [n_stochastic_steps, B, C, h, w]: described to the C ABI as a stack of [B,C,h,w] slabs
        t_nz = L.as_tensor(nz, shape=(nz.shape[0] * nz.shape[1],) + tuple(nz.shape[2:])) if nz is not None else None
        t_c = L.as_tensor(cond) if cond is not None else None
        f_arr = L.tensor_array([L.as_tensor(f) for f in feats])
        L.check(L.lib().mrisr_sampler_run(self._h, C.byref(t_lat), C.byref(t_lr) if t_lr else None,
                                          C.byref(t_nz) if t_nz else None, C.byref(t_e), C.byref(t_c) if t_c else None,
                                          f_arr if feats else None, len(feats), 1 if use_graph else 0, L.stream_ptr()))
        self._keep = keep
        return latents


@torch.no_grad()
def log_validation(unet, controlnet, vae, val_dataloader, noise_scheduler, weight_dtype, accelerator, fixed_embeds,
                   num_inference_steps=20):
    """Drop-in for the reference's validation sampler (res_srdiff.py:35-105): same inputs, same PIL panel out.
    The timestep loop is one fused sampler call; the per-step noise is drawn up front from the same global RNG
    stream, in the same order, as the reference's per-step ``torch.randn_like`` calls."""
    from PIL import Image

    unet.eval()
    if controlnet is not None:
        controlnet.eval()
    dev = accelerator.device
    val_batch = next(iter(val_dataloader))
    hr_raw = val_batch["hr"][0:1].to(dev, dtype=weight_dtype)
    lr_raw = val_batch["lr"][0:1].to(dev, dtype=weight_dtype)
    control_image = prepare_condition_image(lr_raw)
    lr_input = lr_raw.expand(-1, 3, -1, -1) if lr_raw.shape[1] == 1 else lr_raw
    lr_anchor = (vae.encode(lr_input).latent_dist.sample() * vae.config.scaling_factor).to(torch.float32)
    noise_scheduler.set_timesteps(num_inference_steps, device=dev)
    timesteps = noise_scheduler.timesteps
    latents = get_res_shifting_latents(lr_anchor, lr_anchor, timesteps[0], noise_scheduler).contiguous()
    n_noise = sum(1 for i in range(len(timesteps)) if (int(timesteps[i + 1]) if i + 1 < len(timesteps) else 0) > 0)
    step_noise = torch.stack([torch.randn_like(latents) for _ in range(n_noise)]) if n_noise else None
    sampler = Sampler(unet, noise_scheduler, controlnet, kind="resshift")
    sampler.run(latents, fixed_embeds[0:1], lr_latents=lr_anchor, step_noise=step_noise,
                controlnet_cond=control_image if controlnet is not None else None)
    gen_vis = decode_to_vis(latents.to(weight_dtype), vae)
    hr_vis = decode_to_vis(hr_raw, vae, is_latent=False)
    lr_vis = decode_to_vis(lr_raw, vae, is_latent=False)
    return Image.fromarray(np.hstack([lr_vis, gen_vis, hr_vis]))
