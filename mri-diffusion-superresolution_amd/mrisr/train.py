"""LoRA fine-tuning step on the device library.

Mirrors the reference's training cell (notebook ResDif c11:14-41)::

    noise_pred = unet(noisy_latents, timesteps, encoder_hidden_states, ...).sample
    loss = F.mse_loss(noise_pred, noise); accelerator.backward(loss)
    accelerator.clip_grad_norm_(params, 1.0); optimizer.step(); lr_scheduler.step(); optimizer.zero_grad()

Here forward, loss, backward and AdamW all run in ``libmrisr.so``; the only thing this file adds is the one exchange
step of data parallelism: the adapters' gradients live in ONE flat f32 vector, which is all-reduced (RCCL on GPUs,
``torch.distributed`` backend "nccl") once per step.  LoRA r=4 on SD-1.5 is 797,184 floats = 3.19 MB: a single
latency-bound bucket, no overlap machinery needed.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from .dist import BucketedReducer, all_reduce_sum_


def cosine_lr(step: int, base_lr: float, warmup_steps: int, total_steps: int, num_cycles: float = 0.5) -> float:
    """diffusers ``get_cosine_schedule_with_warmup`` (``lr_scheduler_name: "cosine"``, nb ResDif c11:25: half a cosine, since
    ``get_scheduler`` hands ``num_cycles`` only to the with-restarts variant).  Like diffusers the progress is NOT clamped:
    past ``total_steps`` the multiplier follows the cosine on (and is floored at 0)."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    p = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * p)))


# LoRA checkpoint key forms.  In memory (and in ``state_dict()``) the adapters carry peft's keys with the adapter name,
# ``<module>.lora_A.default.weight``.  On disk peft writes them WITHOUT the adapter name under a ``base_model.model.`` prefix
# (``get_peft_model_state_dict`` / ``set_peft_model_state_dict``), diffusers' LoRA loaders use ``unet.<module>.lora_A.weight``.
_KEY_PREFIX = {"peft": "base_model.model.", "diffusers": "unet.", "memory": ""}


def lora_keys_to_disk(sd: Dict[str, torch.Tensor], key_format: str = "peft") -> Dict[str, torch.Tensor]:
    if key_format not in _KEY_PREFIX:
        raise ValueError(f"key_format must be one of {sorted(_KEY_PREFIX)}")
    if key_format == "memory":
        return dict(sd)
    pre = _KEY_PREFIX[key_format]
    return {pre + k.replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B."): v for k, v in sd.items()}


def lora_keys_from_disk(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Any of the three forms -> the in-memory peft keys (other keys pass through)."""
    out = {}
    for k, v in sd.items():
        for pre in ("base_model.model.", "unet."):
            if k.startswith(pre):
                k = k[len(pre):]
                break
        for ab in ("lora_A", "lora_B"):
            if k.endswith(f".{ab}.weight"):
                k = k[:-len(f".{ab}.weight")] + f".{ab}.default.weight"
        out[k] = v
    return out


class _FlatAdamW:
    """clip_grad_norm_ + AdamW on one flat f32 parameter / gradient vector pair (device kernels mrisr_optim_*)."""

    def _init_flat(self, n: int, dev, lr, betas, weight_decay, eps, max_grad_norm, process_group):
        self.lr, self.betas, self.weight_decay, self.eps, self.max_grad_norm = lr, betas, weight_decay, eps, max_grad_norm
        self.group = process_group
        self.theta = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_count = 0

    @property
    def num_trainable(self) -> int:
        return self.theta.numel()

    def zero_grad(self):
        self.grad.zero_()

    def all_reduce_grads(self) -> int:
        """The data-parallel exchange step: SUM over ranks of the flat gradient bucket; returns the world size."""
        return all_reduce_sum_(self.grad, self.group)

    def sumsq(self) -> torch.Tensor:
        """Squared L2 norm of the (reduced) gradient bucket, on the device."""
        self._sumsq.zero_()
        L.check(L.lib().mrisr_optim_sumsq(C.c_void_p(self.grad.data_ptr()), C.c_int64(self.grad.numel()),
                                          C.c_void_p(self._sumsq.data_ptr()), L.stream_ptr()))
        return self._sumsq

    def _adamw(self, world: int, lr: Optional[float], sumsq: Optional[torch.Tensor] = None):
        """``sumsq``: squared norm to clip against (default: this bucket's own; pass the sum over several buckets to clip
        them jointly, as one ``clip_grad_norm_`` over all trainable parameters does)."""
        self.step_count += 1
        ss = self.sumsq() if sumsq is None else sumsq
        L.check(L.lib().mrisr_optim_adamw(C.c_void_p(self.theta.data_ptr()), C.c_void_p(self.grad.data_ptr()),
                                          C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()),
                                          C.c_int64(self.theta.numel()), C.c_void_p(ss.data_ptr()), C.c_float(1.0 / world),
                                          C.c_float(self.max_grad_norm or 0.0), C.c_float(self.lr if lr is None else lr),
                                          C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                          C.c_float(self.weight_decay), C.c_int(self.step_count), L.stream_ptr()))
        self._last_sumsq = ss

    def _adamw_range(self, lo: int, hi: int, world: int, lr: Optional[float], sumsq: torch.Tensor):
        """AdamW on the slice [lo, hi) only (the sharded optimiser of ``joint_step_overlapped``); ``step_count`` is the caller's."""
        es = 4
        L.check(L.lib().mrisr_optim_adamw(C.c_void_p(self.theta.data_ptr() + lo * es), C.c_void_p(self.grad.data_ptr() + lo * es),
                                          C.c_void_p(self.exp_avg.data_ptr() + lo * es), C.c_void_p(self.exp_avg_sq.data_ptr() + lo * es),
                                          C.c_int64(hi - lo), C.c_void_p(sumsq.data_ptr()), C.c_float(1.0 / world),
                                          C.c_float(self.max_grad_norm or 0.0), C.c_float(self.lr if lr is None else lr),
                                          C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                          C.c_float(self.weight_decay), C.c_int(self.step_count), L.stream_ptr()))
        self._last_sumsq = sumsq

    # ---- EMA of the trainable parameters (diffusers EMAModel: use_ema in the reference's training config) ----
    def ema_init(self):
        """EMAModel(parameters): the shadow starts as a copy of the parameters at construction time."""
        self.ema = self.theta.clone()
        self.ema_steps = 0

    @staticmethod
    def ema_decay_at(optimization_step: int, decay: float = 0.9999, min_decay: float = 0.0, update_after_step: int = 0,
                     use_ema_warmup: bool = False, inv_gamma: float = 1.0, power: float = 2.0 / 3.0) -> float:
        """diffusers ``EMAModel.get_decay``: 0 until ``update_after_step``, then (1+s)/(10+s) (or the warm-up curve
        1-(1+s/inv_gamma)^-power), clamped to [min_decay, decay]."""
        step = max(0, optimization_step - update_after_step - 1)
        if step <= 0:
            return 0.0
        cur = 1.0 - (1.0 + step / inv_gamma) ** -power if use_ema_warmup else (1.0 + step) / (10.0 + step)
        return max(min(cur, decay), min_decay)

    def ema_step(self, decay: float = 0.9999, **schedule):
        """One ``EMAModel.step``: shadow -= (1 - d) (shadow - theta) with d = ``ema_decay_at(step count)``; ``decay`` is the
        ceiling of the schedule, as in diffusers (NOT a constant rate)."""
        if getattr(self, "ema", None) is None:
            self.ema_init()
        self.ema_steps = getattr(self, "ema_steps", 0) + 1
        d = self.ema_decay_at(self.ema_steps, decay, **schedule)
        L.check(L.lib().mrisr_optim_ema(C.c_void_p(self.ema.data_ptr()), C.c_void_p(self.theta.data_ptr()), C.c_int64(self.theta.numel()),
                                        C.c_float(d), L.stream_ptr()))
        return d

    # ---- checkpoints: parameters as a safetensors file in peft's ON-DISK key form by default (adapter name stripped,
    #      ``base_model.model.`` prefix: what ``set_peft_model_state_dict`` / ``PeftModel.from_pretrained`` read), or diffusers'
    #      ``unet.`` form; conv parameters of the T2I-Adapter keep the reference module's own keys.  Optimiser state as flat
    #      vectors next to it. ----
    def save_checkpoint(self, path: str, use_ema: bool = False, key_format: str = "peft"):
        from safetensors.torch import save_file
        flat = self.ema if (use_ema and getattr(self, "ema", None) is not None) else self.theta
        sd = {k: v.contiguous().cpu() for k, v in self._views(flat).items()}
        save_file(lora_keys_to_disk(sd, key_format if any(".lora_" in k for k in sd) else "memory"), path)
        torch.save({"step": self.step_count, "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(),
                    "ema": None if getattr(self, "ema", None) is None else self.ema.cpu(),
                    "ema_steps": getattr(self, "ema_steps", 0)}, path + ".optim.pt")

    def load_checkpoint(self, path: str):
        from safetensors.torch import load_file
        sd = lora_keys_from_disk(load_file(path))
        missing = [k for k in self._views(self.theta) if k not in sd]
        if missing:
            raise KeyError(f"checkpoint {path} lacks {len(missing)} trainable tensors, e.g. {missing[0]}")
        self.load_state_dict(sd)
        if os.path.exists(path + ".optim.pt"):
            st = torch.load(path + ".optim.pt", map_location="cpu")
            self.step_count = int(st["step"])
            self.exp_avg.copy_(st["exp_avg"])
            self.exp_avg_sq.copy_(st["exp_avg_sq"])
            self.ema = None if st["ema"] is None else st["ema"].to(self.theta.device)
            self.ema_steps = int(st.get("ema_steps", 0))

    def grad_norm(self, world: int = 1) -> float:
        """Global L2 norm of the (averaged) gradient as used by the last ``optimizer_step``."""
        return float(self._last_sumsq.sqrt().item()) / world


class LoRATrainer(_FlatAdamW):
    """Owns the flat trainable / gradient / AdamW-moment vectors of a ``UNet2DConditionModel`` created with
    ``lora_rank > 0, lora_fused=True`` and drives one optimisation step."""

    def __init__(self, unet, lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.999), weight_decay: float = 1e-2,
                 eps: float = 1e-8, max_grad_norm: float = 1.0, process_group=None):
        if not getattr(unet, "_finalized", False):
            raise L.MrisrError("load_state_dict() first")
        self.unet = unet
        lib = L.lib()
        lib.mrisr_train_num_trainable.restype = C.c_int64
        lib.mrisr_train_num_trainable.argtypes = [C.c_void_p]
        lib.mrisr_train_num_tensors.argtypes = [C.c_void_p]
        L.check(lib.mrisr_train_prepare(unet._h, L.stream_ptr()))
        n = int(lib.mrisr_train_num_trainable(unet._h))
        dev = unet.device
        self._init_flat(n, dev, lr, betas, weight_decay, eps, max_grad_norm, process_group)
        self._loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.layout: List[Tuple[str, int, Tuple[int, int]]] = []
        for i in range(int(lib.mrisr_train_num_tensors(unet._h))):
            key, off, shp = C.c_char_p(), C.c_int64(), (C.c_int64 * 2)()
            L.check(lib.mrisr_train_tensor_info(unet._h, i, C.byref(key), C.byref(off), shp))
            self.layout.append((key.value.decode(), int(off.value), (int(shp[0]), int(shp[1]))))
        L.check(lib.mrisr_train_bind(unet._h, C.c_void_p(self.theta.data_ptr()), C.c_void_p(self.grad.data_ptr()), 1,
                                     L.stream_ptr()))
        import weakref
        unet._trainer = weakref.ref(self)  # unet.state_dict() reads the adapters' CURRENT values from this trainer

    # ---- views ----
    def _views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {k: flat[o:o + r * c].view(r, c) for k, o, (r, c) in self.layout}

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """The adapters under their peft keys (what ``unet.save_attn_procs`` / ``get_peft_model_state_dict`` would hold)."""
        return {k: v.clone() for k, v in self._views(self.theta).items()}

    def gradients(self) -> Dict[str, torch.Tensor]:
        return self._views(self.grad)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        views = self._views(self.theta)
        for k, v in lora_keys_from_disk(sd).items():
            if k in views:
                views[k].copy_(v.to(self.theta.device, torch.float32))
        L.check(L.lib().mrisr_train_refresh(self.unet._h, L.stream_ptr()))

    # ---- the step, in the reference's order ----
    def forward_backward(self, noisy_latents: torch.Tensor, timesteps, encoder_hidden_states: torch.Tensor,
                         target: torch.Tensor, down_intrablock_additional_residuals: Optional[Sequence[torch.Tensor]] = None,
                         return_pred: bool = False, feature_grads: Optional[Sequence[torch.Tensor]] = None,
                         down_block_additional_residuals: Optional[Sequence[torch.Tensor]] = None,
                         mid_block_additional_residual: Optional[torch.Tensor] = None, residual_grads=None):
        """loss = mse(unet(noisy_latents, timesteps, ehs [, ControlNet residuals] [, adapter features]), target); adds
        d(loss)/d(adapters) to ``self.grad``.  ``residual_grads = (list of 12 tensors, mid tensor)``: d(loss)/d(each ControlNet
        residual) is written there (the seeds of ``ControlNetTrainer.backward``).  The UNet may be frozen (``lora_rank=0``).
        Returns the loss as a device scalar (and eps_hat when ``return_pred``)."""
        u = self.unet
        x = noisy_latents.to(u.device).contiguous()
        B = x.shape[0]
        t = u._timestep(timesteps, B)
        ehs = encoder_hidden_states.to(u.device).contiguous()
        tgt = target.to(u.device, torch.float32).contiguous()
        intra = [r.to(u.device).contiguous() for r in (down_intrablock_additional_residuals or [])]
        i_arr = L.tensor_array([L.as_tensor(r) for r in intra])
        pred = torch.empty((B, u.config.out_channels, x.shape[2], x.shape[3]), dtype=torch.float32, device=u.device) if return_pred else None
        t_x, t_t, t_e, t_g = L.as_tensor(x), L.as_tensor(t), L.as_tensor(ehs), L.as_tensor(tgt)
        t_p = L.as_tensor(pred) if pred is not None else None
        # optional: d(loss)/d(adapter features), written into the caller's tensors for the adapter's own backward
        g_arr = L.tensor_array([L.as_tensor(f) for f in (feature_grads or [])])
        L.check(L.lib().mrisr_train_set_intrablock_grads(u._h, g_arr if feature_grads else None, len(feature_grads or [])))
        down = [r.to(u.device).contiguous() for r in (down_block_additional_residuals or [])]
        mid = mid_block_additional_residual.to(u.device).contiguous() if mid_block_additional_residual is not None else None
        if residual_grads is not None and (len(residual_grads[0]) != len(down) or (residual_grads[1] is None) != (mid is None)):
            raise ValueError("residual_grads must mirror the residuals: (list like down_block_additional_residuals, tensor like mid)")
        d_arr = L.tensor_array([L.as_tensor(r) for r in down])
        dg_arr = L.tensor_array([L.as_tensor(r) for r in residual_grads[0]]) if residual_grads is not None else None
        t_mid = L.as_tensor(mid) if mid is not None else None
        t_dmid = L.as_tensor(residual_grads[1]) if (residual_grads is not None and mid is not None) else None
        L.check(L.lib().mrisr_train_set_controlnet_residuals(u._h, d_arr if down else None, dg_arr if (down and dg_arr is not None) else None, len(down),
                                                             C.byref(t_mid) if t_mid else None, C.byref(t_dmid) if t_dmid else None))
        L.check(L.lib().mrisr_train_step(u._h, C.byref(t_x), C.byref(t_t), C.byref(t_e), i_arr if intra else None, len(intra),
                                         C.byref(t_g), C.c_void_p(self._loss.data_ptr()), C.byref(t_p) if t_p else None,
                                         L.stream_ptr()))
        loss = self._loss.clone()[0]
        return (loss, pred) if return_pred else loss

    def optimizer_step(self, world: int = 1, lr: Optional[float] = None, sumsq: Optional[torch.Tensor] = None):
        """clip_grad_norm_(max_grad_norm) + AdamW on the (already reduced) gradients, then re-pack the adapters."""
        self._adamw(world, lr, sumsq)
        L.check(L.lib().mrisr_train_refresh(self.unet._h, L.stream_ptr()))

    def step(self, noisy_latents, timesteps, encoder_hidden_states, target, down_intrablock_additional_residuals=None,
             lr: Optional[float] = None):
        self.zero_grad()
        loss = self.forward_backward(noisy_latents, timesteps, encoder_hidden_states, target, down_intrablock_additional_residuals)
        world = self.all_reduce_grads()
        self.optimizer_step(world, lr)
        return loss


def _numel(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


def controlnet_bucket_ranges(layout: Sequence[Tuple[str, int, int]]) -> List[Tuple[str, int, int]]:
    """``layout``: (key, offset, numel) of a ControlNet's flat parameter vector (offsets in sorted-key order, as the library lays
    them out).  Returns (group, lo, hi) in backward-finalisation order; the groups tile [0, total) exactly."""
    def group(k: str) -> str:
        if k.startswith("controlnet_down_blocks.") or k.startswith("controlnet_mid_block."):
            return "zero_convs"
        if k.startswith("down_blocks."):
            return "down_blocks." + k.split(".")[1]
        for g in ("mid_block", "conv_in", "controlnet_cond_embedding", "time_embedding"):
            if k.startswith(g + "."):
                return g
        raise KeyError(f"not a ControlNet parameter: {k}")
    spans: Dict[str, List[int]] = {}
    for k, o, n in layout:
        g = group(k)
        lo, hi = spans.get(g, [o, o + n])
        spans[g] = [min(lo, o), max(hi, o + n)]
    # sorted keys put controlnet_down_blocks.* and controlnet_mid_block.* side by side, and every other group in one run
    total = sum(n for _, _, n in layout)
    assert sum(hi - lo for lo, hi in spans.values()) == total, "groups must be contiguous runs of the sorted-key layout"
    levels = sorted((g for g in spans if g.startswith("down_blocks.")), key=lambda g: -int(g.split(".")[1]))
    order = ["zero_convs", "mid_block"] + levels + ["conv_in", "controlnet_cond_embedding", "time_embedding"]
    return [(g, spans[g][0], spans[g][1]) for g in order if g in spans]


class ControlNetTrainer(_FlatAdamW):
    """A ``ControlNetModel`` with its own parameters trainable (the reference itself only runs a ControlNet for
    inference, res_srdiff.py:65-70).  All raw tensors live in one flat f32 vector (``self.theta``; ``self.layout`` maps state-dict keys
    to offsets); ``self.frozen`` lists the tensors this build does not differentiate (norm affine, time embedding, condition embedding):
    their gradient stays zero and ``optimizer_step`` masks them out of the weight decay as well.

        down, mid = cn.forward(x_t, t, ehs, cond)                       # recorded forward
        dg = ([torch.zeros_like(d) for d in down], torch.zeros_like(mid))
        loss = unet_trainer.forward_backward(x_t, t, ehs, target, down_block_additional_residuals=down,
                                             mid_block_additional_residual=mid, residual_grads=dg)
        cn.backward(*dg); cn.optimizer_step()
    """

    def __init__(self, controlnet, lr: float = 1e-5, betas: Tuple[float, float] = (0.9, 0.999), weight_decay: float = 1e-2,
                 eps: float = 1e-8, max_grad_norm: float = 1.0, process_group=None, conditioning_scale: float = 1.0):
        if not getattr(controlnet, "_finalized", False):
            raise L.MrisrError("load_state_dict() first")
        self.controlnet, self.scale = controlnet, float(conditioning_scale)
        lib = L.lib()
        lib.mrisr_controlnet_train_num_trainable.restype = C.c_int64
        lib.mrisr_controlnet_train_num_trainable.argtypes = [C.c_void_p]
        lib.mrisr_controlnet_train_num_tensors.argtypes = [C.c_void_p]
        L.check(lib.mrisr_controlnet_train_prepare(controlnet._h, L.stream_ptr()))
        n = int(lib.mrisr_controlnet_train_num_trainable(controlnet._h))
        self._init_flat(n, controlnet.device, lr, betas, weight_decay, eps, max_grad_norm, process_group)
        shapes = {k: tuple(v.shape) for k, v in controlnet._params.items()}
        self.layout: List[Tuple[str, int, Tuple[int, ...]]] = []
        self.frozen: List[str] = []
        for i in range(int(lib.mrisr_controlnet_train_num_tensors(controlnet._h))):
            key, off, num, ok = C.c_char_p(), C.c_int64(), C.c_int64(), C.c_int()
            L.check(lib.mrisr_controlnet_train_tensor_info(controlnet._h, i, C.byref(key), C.byref(off), C.byref(num), C.byref(ok)))
            k = key.value.decode()
            self.layout.append((k, int(off.value), shapes.get(k, (int(num.value),))))
            if not ok.value:
                self.frozen.append(k)
        L.check(lib.mrisr_controlnet_train_bind(controlnet._h, C.c_void_p(self.theta.data_ptr()), C.c_void_p(self.grad.data_ptr()), 1, L.stream_ptr()))

    def _views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {}
        for k, o, shp in self.layout:
            n = 1
            for d in shp:
                n *= d
            out[k] = flat[o:o + n].view(*shp) if shp else flat[o:o + 1].view(())
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.clone() for k, v in self._views(self.theta).items()}

    def gradients(self) -> Dict[str, torch.Tensor]:
        return self._views(self.grad)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        views = self._views(self.theta)
        for k, v in sd.items():
            if k in views:
                views[k].copy_(v.to(self.theta.device, torch.float32))
        L.check(L.lib().mrisr_controlnet_train_refresh(self.controlnet._h, L.stream_ptr()))

    def forward(self, sample: torch.Tensor, timesteps, encoder_hidden_states: torch.Tensor, controlnet_cond: torch.Tensor):
        """``controlnet(sample, t, encoder_hidden_states=..., controlnet_cond=..., return_dict=False)`` with the forward recorded for
        ONE following ``backward``.  Returns (list of 12 residuals, mid residual), f32 NCHW."""
        cn = self.controlnet
        x = sample.to(cn.device).contiguous()
        B, _, h, w = x.shape
        t = cn._timestep(timesteps, B)
        ehs = encoder_hidden_states.to(cn.device).contiguous()
        cond = controlnet_cond.to(cn.device).contiguous()
        shapes = cn._skip_shapes(B, h, w)
        outs = [torch.empty(s, dtype=torch.float32, device=cn.device) for s in shapes]
        o_arr = L.tensor_array([L.as_tensor(o) for o in outs[:-1]])
        t_x, t_t, t_e, t_c, t_m = L.as_tensor(x), L.as_tensor(t), L.as_tensor(ehs), L.as_tensor(cond), L.as_tensor(outs[-1])
        L.check(L.lib().mrisr_controlnet_train_forward(cn._h, C.byref(t_x), C.byref(t_t), C.byref(t_e), C.byref(t_c), C.c_float(self.scale),
                                                       o_arr, len(outs) - 1, C.byref(t_m), L.stream_ptr()))
        return outs[:-1], outs[-1]

    def backward(self, d_down: Sequence[torch.Tensor], d_mid: torch.Tensor):
        """Adds d(loss)/d(parameter) to ``self.grad`` given d(loss)/d(residual) (what ``LoRATrainer.forward_backward(...,
        residual_grads=...)`` wrote)."""
        cn = self.controlnet
        g = [d.to(cn.device).contiguous() for d in d_down]
        gm = d_mid.to(cn.device).contiguous()
        arr = L.tensor_array([L.as_tensor(d) for d in g])
        t_m = L.as_tensor(gm)
        L.check(L.lib().mrisr_controlnet_train_backward(cn._h, arr, len(g), C.byref(t_m), C.c_float(self.scale), L.stream_ptr()))

    def bucket_ranges(self) -> List[Tuple[str, int, int]]:
        """Contiguous ranges [lo, hi) of the flat vector in the order in which the backward FINALISES them - what
        ``mrisr.dist.BucketedReducer`` takes: the zero convs (first kernels of the backward), the mid block, the down blocks from
        the deepest level up, conv_in, the condition embedding, and the time embedding last (it collects from every ResnetBlock)."""
        return controlnet_bucket_ranges([(k, o, _numel(shp)) for k, o, shp in self.layout])

    def step(self, unet_trainer: "LoRATrainer", noisy_latents, timesteps, encoder_hidden_states, target, controlnet_cond,
             lr: Optional[float] = None):
        """One whole training step of the ControlNet configuration: recorded ControlNet forward, the UNet step with its residuals
        (frozen UNet, or LoRA trained alongside: then ``unet_trainer``'s own gradients are filled too and the caller steps it),
        ControlNet backward, all-reduce of the 1.45 GB bucket, clip + AdamW, in-place re-pack.  Returns the loss."""
        self.zero_grad()
        down, mid = self.forward(noisy_latents, timesteps, encoder_hidden_states, controlnet_cond)
        dg = ([torch.zeros_like(d) for d in down], torch.zeros_like(mid))
        loss = unet_trainer.forward_backward(noisy_latents, timesteps, encoder_hidden_states, target, down_block_additional_residuals=down,
                                             mid_block_additional_residual=mid, residual_grads=dg)
        self.backward(*dg)
        world = self.all_reduce_grads()
        self.optimizer_step(world, lr)
        return loss

    def optimizer_step(self, world: int = 1, lr: Optional[float] = None, sumsq: Optional[torch.Tensor] = None):
        """clip + AdamW on the flat vectors, then the new values are re-packed into the handle (forward and dgrad weight copies)."""
        if self.frozen:  # no gradient -> no update, weight decay included: restore those slices after the step
            views = self._views(self.theta)
            keep = {k: views[k].clone() for k in self.frozen}
        self._adamw(world, lr, sumsq)
        if self.frozen:
            views = self._views(self.theta)
            for k, v in keep.items():
                views[k].copy_(v)
        L.check(L.lib().mrisr_controlnet_train_refresh(self.controlnet._h, L.stream_ptr()))


class AdapterTrainer(_FlatAdamW):
    """The reference's T2I-Adapter (``Adapter_XL``, src/adapters/modules.py:114-157) as a trainable module: every conv
    weight / bias lives in one flat f32 vector (state-dict keys and shapes via ``layout``), ``forward`` is the adapter
    forward that also keeps what ``backward`` needs, ``backward`` consumes the feature gradients the UNet step produced."""

    def __init__(self, adapter, lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.999), weight_decay: float = 1e-2,
                 eps: float = 1e-8, max_grad_norm: float = 1.0, process_group=None):
        if not getattr(adapter, "_finalized", False):
            raise L.MrisrError("load_state_dict() first")
        self.adapter = adapter
        lib = L.lib()
        lib.mrisr_adapter_train_num_trainable.restype = C.c_int64
        lib.mrisr_adapter_train_num_trainable.argtypes = [C.c_void_p]
        lib.mrisr_adapter_train_num_tensors.argtypes = [C.c_void_p]
        L.check(lib.mrisr_adapter_train_prepare(adapter._h, L.stream_ptr()))
        self._init_flat(int(lib.mrisr_adapter_train_num_trainable(adapter._h)), adapter.device, lr, betas, weight_decay, eps,
                        max_grad_norm, process_group)
        self.layout: List[Tuple[str, int, Tuple[int, ...]]] = []
        for i in range(int(lib.mrisr_adapter_train_num_tensors(adapter._h))):
            key, off, shp, nd = C.c_char_p(), C.c_int64(), (C.c_int64 * 4)(), C.c_int()
            L.check(lib.mrisr_adapter_train_tensor_info(adapter._h, i, C.byref(key), C.byref(off), shp, C.byref(nd)))
            self.layout.append((key.value.decode(), int(off.value), tuple(int(shp[k]) for k in range(nd.value))))
        L.check(lib.mrisr_adapter_train_bind(adapter._h, C.c_void_p(self.theta.data_ptr()), C.c_void_p(self.grad.data_ptr()), 1,
                                             L.stream_ptr()))

    def _views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {k: flat[o:o + math.prod(s)].view(s) for k, o, s in self.layout}

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.clone() for k, v in self._views(self.theta).items()}

    def gradients(self) -> Dict[str, torch.Tensor]:
        return self._views(self.grad)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        views = self._views(self.theta)
        for k, v in sd.items():
            if k in views:
                views[k].copy_(v.to(self.theta.device, torch.float32))
        L.check(L.lib().mrisr_adapter_train_refresh(self.adapter._h, L.stream_ptr()))

    def forward(self, x: torch.Tensor):
        """Adapter features for ``down_intrablock_additional_residuals`` (kept for ``backward``)."""
        self._feats = self.adapter(x)
        return self._feats

    def new_feature_grads(self):
        """Zero tensors shaped like the last features, to hand to ``LoRATrainer.forward_backward(feature_grads=...)``."""
        return [torch.zeros_like(f) for f in self._feats]

    def backward(self, feature_grads: Sequence[torch.Tensor]):
        g = [f.to(self.adapter.device).contiguous() for f in feature_grads]
        arr = L.tensor_array([L.as_tensor(f) for f in g])
        L.check(L.lib().mrisr_adapter_backward(self.adapter._h, arr, len(g), L.stream_ptr()))

    # ---- the same pass, one level at a time (top level first): lets the exchange of a level's gradients overlap the rest ----
    @property
    def num_levels(self) -> int:
        return len(self.adapter.channels)

    def level_range(self, level: int) -> Tuple[int, int]:
        """[lo, hi) of the flat trainable / gradient vector holding this level's convolutions (level 0: incl. conv_in)."""
        off, n = C.c_int64(), C.c_int64()
        L.check(L.lib().mrisr_adapter_train_level_range(self.adapter._h, int(level), C.byref(off), C.byref(n)))
        return int(off.value), int(off.value + n.value)

    def backward_level(self, feature_grads: Sequence[torch.Tensor], level: int):
        g = [f.to(self.adapter.device).contiguous() for f in feature_grads]
        arr = L.tensor_array([L.as_tensor(f) for f in g])
        L.check(L.lib().mrisr_adapter_backward_level(self.adapter._h, arr, len(g), int(level), L.stream_ptr()))

    def optimizer_step(self, world: int = 1, lr: Optional[float] = None, sumsq: Optional[torch.Tensor] = None):
        self._adamw(world, lr, sumsq)
        L.check(L.lib().mrisr_adapter_train_refresh(self.adapter._h, L.stream_ptr()))


def joint_step_overlapped(lora: LoRATrainer, adapter: AdapterTrainer, noisy_latents, timesteps, encoder_hidden_states, target,
                          adapter_input, lr: Optional[float] = None, mode: str = "all_reduce"):
    """``joint_step`` with the large-bucket exchange of SURVEY.md 8e: the adapter's 935 MB gradient vector is reduced level by
    level (buckets in backward order: level 3 first) while the lower levels are still being differentiated, instead of one
    all-reduce after the whole backward; ``mode="reduce_scatter"`` additionally shards the adapter's AdamW step over the ranks
    (``BucketedReducer``).  Same arithmetic as ``joint_step``: the clip uses the global norm over both parameter sets."""
    lora.zero_grad()
    adapter.zero_grad()
    feats = adapter.forward(adapter_input)
    fg = adapter.new_feature_grads()
    loss = lora.forward_backward(noisy_latents, timesteps, encoder_hidden_states, target, down_intrablock_additional_residuals=feats,
                                 feature_grads=fg)
    levels = list(range(adapter.num_levels - 1, -1, -1))
    red = BucketedReducer(adapter.grad, [adapter.level_range(l) for l in levels], adapter.group, mode)
    # the LoRA bucket (3.19 MB) is final already: its collective goes first and hides behind the adapter's backward too
    lred = BucketedReducer(lora.grad, [(0, lora.grad.numel())], lora.group, "all_reduce")
    lred.reduce(0)
    for i, l in enumerate(levels):
        adapter.backward_level(fg, l)
        red.reduce(i)
    world = red.wait()
    lred.wait()
    if mode == "all_reduce" or world == 1:
        total = lora.sumsq().clone() + adapter.sumsq()
        lora.optimizer_step(world, lr, sumsq=total)
        adapter.optimizer_step(world, lr, sumsq=total)
        return loss
    # reduce_scatter: this rank holds the summed gradient of its shards only - its share of the squared norm, one scalar
    # all-reduce, then AdamW on the shards and an all-gather of the updated parameters
    import torch.distributed as dist
    part = torch.zeros(1, dtype=torch.float32, device=adapter.grad.device)
    for i in range(len(levels)):
        lo, hi = red.shard(i)
        part += adapter.grad[lo:hi].double().pow(2).sum().float()
    dist.all_reduce(part, group=adapter.group)
    total = lora.sumsq().clone() + part
    lora.optimizer_step(world, lr, sumsq=total)
    adapter.step_count += 1
    for i in range(len(levels)):
        lo, hi = red.shard(i)
        if hi > lo:
            adapter._adamw_range(lo, hi, world, lr, total)
    red.all_gather_params(adapter.theta)
    L.check(L.lib().mrisr_adapter_train_refresh(adapter.adapter._h, L.stream_ptr()))
    return loss


def joint_step(lora: LoRATrainer, adapter: AdapterTrainer, noisy_latents, timesteps, encoder_hidden_states, target, adapter_input,
               lr: Optional[float] = None):
    """One optimisation step of BASELINE config 3: adapter forward -> UNet forward / loss / backward (LoRA gradients +
    feature gradients) -> adapter backward -> all-reduce of both gradient buckets -> ONE global clip over all trainable
    parameters -> AdamW on both."""
    lora.zero_grad()
    adapter.zero_grad()
    feats = adapter.forward(adapter_input)
    fg = adapter.new_feature_grads()
    loss = lora.forward_backward(noisy_latents, timesteps, encoder_hidden_states, target, down_intrablock_additional_residuals=feats,
                                 feature_grads=fg)
    adapter.backward(fg)
    world = lora.all_reduce_grads()
    adapter.all_reduce_grads()
    total = lora.sumsq().clone() + adapter.sumsq()   # clip_grad_norm_ over the union of both parameter sets
    lora.optimizer_step(world, lr, sumsq=total)
    adapter.optimizer_step(world, lr, sumsq=total)
    return loss
