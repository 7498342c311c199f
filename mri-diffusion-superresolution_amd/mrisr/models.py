"""Host-side mirror of the model objects the reference passes around (duck-typed):

* ``UNet2DConditionModel``  - call surface of diffusers' class as used at reference
  ``src/adapters/res_srdiff.py:73-78``:  ``unet(latents, t, encoder_hidden_states=..., down_block_additional_residuals=...,
  mid_block_additional_residual=...).sample``  (+ ``down_intrablock_additional_residuals`` for T2I-Adapter features).
* ``ControlNetModel``       - ``controlnet(latents, t, encoder_hidden_states=..., controlnet_cond=..., return_dict=False)``
  -> ``(down_res, mid_res)``  (``res_srdiff.py:65-70``).
* ``Adapter_XL``            - the reference's own T2I-Adapter (``src/adapters/modules.py:114-157``), ``sk=True`` semantics.

All arithmetic runs in libmrisr.so (HIP, gfx950); these classes only hold parameters (diffusers / peft / reference
state-dict key names) and marshal device pointers.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L


@dataclass
class UNetConfig:
    """diffusers config keys the path depends on (SD-1.5 defaults)."""
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                         "DownBlock2D")
    layers_per_block: int = 2
    attention_head_dim: int = 8  # number of heads (diffusers naming quirk)
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    conditioning_channels: int = 3
    conditioning_embedding_out_channels: Tuple[int, ...] = (16, 32, 96, 256)

    @classmethod
    def from_oracle_like(cls, cfg) -> "UNetConfig":
        """Build from any object with the oracle's/diffusers' field names (attn_levels or down_block_types)."""
        if hasattr(cfg, "attn_levels"):
            types = tuple("CrossAttnDownBlock2D" if a else "DownBlock2D" for a in cfg.attn_levels)
        else:
            types = tuple(cfg.down_block_types)
        heads = getattr(cfg, "num_heads", None) or getattr(cfg, "attention_head_dim")
        return cls(in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                   block_out_channels=tuple(cfg.block_out_channels), down_block_types=types,
                   layers_per_block=cfg.layers_per_block, attention_head_dim=heads,
                   cross_attention_dim=cfg.cross_attention_dim, norm_num_groups=cfg.norm_num_groups,
                   norm_eps=cfg.norm_eps,
                   conditioning_channels=getattr(cfg, "cond_channels", 3),
                   conditioning_embedding_out_channels=tuple(getattr(cfg, "cond_embed_channels", (16, 32, 96, 256))))


class _Output:
    def __init__(self, sample):
        self.sample = sample


class _ControlNetOutput:
    def __init__(self, down, mid):
        self.down_block_res_samples = down
        self.mid_block_res_sample = mid


def _c_cfg(cfg: UNetConfig, compute_dtype, lora_rank, lora_fused, flash, fp8=False, fp8_attention=False, fp8_train=False) -> L.UNetCfg:
    c = L.UNetCfg()
    c.in_channels, c.out_channels = cfg.in_channels, cfg.out_channels
    c.num_levels = len(cfg.block_out_channels)
    for i, ch in enumerate(cfg.block_out_channels):
        c.block_out_channels[i] = ch
        c.attn_levels[i] = 1 if cfg.down_block_types[i].startswith("CrossAttn") else 0
    c.layers_per_block = cfg.layers_per_block
    c.num_heads = cfg.attention_head_dim
    c.cross_attention_dim = cfg.cross_attention_dim
    c.norm_num_groups = cfg.norm_num_groups
    c.norm_eps = cfg.norm_eps
    c.cond_channels = cfg.conditioning_channels
    for i, ch in enumerate(cfg.conditioning_embedding_out_channels):
        c.cond_embed_channels[i] = ch
    c.compute_dtype = L.dtype_id(compute_dtype)
    c.lora_rank = lora_rank
    c.lora_fused = 1 if lora_fused else 0
    c.flash_attention = 1 if flash else 0
    c.fp8_linears = 2 if fp8 == "all" else (1 if fp8 else 0)  # True: the K = 320 projections; "all": K = 320 and 640
    c.fp8_attention = 1 if fp8_attention else 0
    c.fp8_train = 1 if fp8_train else 0
    return c


class _DeviceModel:
    """Shared plumbing: handle lifetime, state-dict <-> C ABI, timestep marshalling."""
    _create = None

    def __init__(self, config=None, compute_dtype="bf16", lora_rank: int = 0, lora_alpha: Optional[float] = None,
                 lora_fused: bool = True, flash_attention: bool = True, device="cuda", fp8=False, fp8_attention: bool = False,
                 fp8_train: bool = False):
        """``fp8`` / ``fp8_attention`` / ``fp8_train``: BASELINE configs[4] - OCP e4m3 operands on the fp8 MFMA for the K = 320 (``"all"``:
        and 640) projections incl. the LoRA targets, for Q K^T / P V of every attention, and for the FORWARD of the training step
        (backward in bf16, straight through).  Modes of the bf16 engine; off by default."""
        if not torch.cuda.is_available():
            raise L.MrisrError("mrisr needs an AMD GPU (gfx950); there is no CPU fallback")
        cfg = config if isinstance(config, UNetConfig) else (UNetConfig() if config is None else UNetConfig.from_oracle_like(config))
        self.config = cfg
        self.device = torch.device(device)
        self.compute_dtype = L.torch_dtype(L.dtype_id(compute_dtype))
        self.lora_rank = lora_rank
        self.lora_scale = (lora_alpha / lora_rank) if (lora_rank and lora_alpha is not None) else 1.0
        self._params: Dict[str, torch.Tensor] = {}
        self._h = C.c_void_p()
        if (fp8 or fp8_attention or fp8_train) and L.dtype_id(compute_dtype) != L.MRISR_BF16:
            raise ValueError("fp8 projections / attention are modes of the bf16 engine")
        if fp8_attention and not flash_attention:
            raise ValueError("fp8 attention is a mode of the flash kernel")
        if fp8_train and not (fp8 or fp8_attention):
            raise ValueError("fp8_train selects the fp8 forward for training: enable fp8 and / or fp8_attention too")
        self.fp8, self.fp8_attention, self.fp8_train = bool(fp8), bool(fp8_attention), bool(fp8_train)
        self._ccfg = _c_cfg(cfg, compute_dtype, lora_rank, lora_fused, flash_attention, fp8, fp8_attention, fp8_train)
        L.check(getattr(L.lib(), self._create)(C.byref(self._ccfg), C.byref(self._h)))
        self._finalized = False
        self.training = False

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                L.lib().mrisr_model_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # ---- nn.Module-like surface ----
    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def to(self, *args, **kwargs):
        return self

    def requires_grad_(self, flag: bool = True):
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        sd = dict(self._params)
        tr = getattr(self, "_trainer", None)
        tr = tr() if tr is not None else None
        if tr is not None:  # a LoRATrainer owns the adapters now: report their trained values, not the loaded ones
            for k, v in tr.state_dict().items():
                sd[k] = v.to(sd[k].dtype) if k in sd else v
        return sd

    def parameters(self):
        return iter(self._params.values())

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """diffusers key names; peft LoRA keys ``<module>.lora_{A,B}.default.weight``."""
        fn = L.lib().mrisr_model_set_param
        for k, v in sd.items():
            # LoRA keys in any of peft's / diffusers' on-disk forms are accepted (adapter name stripped, "base_model.model." or
            # "unet." prefix); everything else must be a diffusers UNet key as is
            if ".lora_A." in k or ".lora_B." in k:
                from .train import lora_keys_from_disk
                k = next(iter(lora_keys_from_disk({k: None})))
            self._params[k] = v
            L.push_param(fn, self._h, k, v)
        L.check(L.lib().mrisr_model_set_lora_scale(self._h, C.c_float(self.lora_scale)))
        L.check(L.lib().mrisr_model_finalize(self._h, L.stream_ptr()))  # raises on missing keys
        self._finalized = True
        return self

    @property
    def num_parameters(self) -> int:
        return int(L.lib().mrisr_model_num_params(self._h))

    @property
    def workspace_bytes(self) -> int:
        return int(L.lib().mrisr_model_workspace_bytes(self._h))

    def _timestep(self, t, batch: int) -> torch.Tensor:
        if not torch.is_tensor(t):
            t = torch.tensor(int(t), dtype=torch.int64)
        t = t.to(device=self.device, dtype=torch.int64)
        if t.ndim > 1 or (t.ndim == 1 and t.shape[0] not in (1, batch)):
            raise ValueError(f"timestep must be 0-dim or [B]; got {tuple(t.shape)}")
        return t.contiguous()

    def _skip_shapes(self, B: int, h: int, w: int) -> List[Tuple[int, int, int, int]]:
        n = L.lib().mrisr_model_num_skips(self._h)
        out = []
        for k in range(n + 1):
            s = (C.c_int64 * 4)()
            L.check(L.lib().mrisr_model_skip_shape(self._h, k, B, h, w, s))
            out.append(tuple(int(x) for x in s))
        return out


class UNet2DConditionModel(_DeviceModel):
    _create = "mrisr_unet_create"

    def __call__(self, sample: torch.Tensor, timestep, encoder_hidden_states: Optional[torch.Tensor] = None,
                 down_block_additional_residuals: Optional[Sequence[torch.Tensor]] = None,
                 mid_block_additional_residual: Optional[torch.Tensor] = None,
                 down_intrablock_additional_residuals: Optional[Sequence[torch.Tensor]] = None,
                 return_dict: bool = True, **_unused):
        if not self._finalized:
            raise L.MrisrError("load_state_dict() first")
        if sample.ndim != 4 or sample.shape[1] != self.config.in_channels:
            raise ValueError(f"sample must be [B,{self.config.in_channels},h,w]; got {tuple(sample.shape)}")
        sample = sample.to(self.device).contiguous()
        B = sample.shape[0]
        t = self._timestep(timestep, B)
        ehs = encoder_hidden_states.to(self.device).contiguous() if encoder_hidden_states is not None else None
        down = [r.to(self.device).contiguous() for r in (down_block_additional_residuals or [])]
        intra = [r.to(self.device).contiguous() for r in (down_intrablock_additional_residuals or [])]
        mid = mid_block_additional_residual.to(self.device).contiguous() if mid_block_additional_residual is not None else None
        out = torch.empty((B, self.config.out_channels, sample.shape[2], sample.shape[3]), dtype=sample.dtype,
                          device=self.device)
        d_arr = L.tensor_array([L.as_tensor(r) for r in down])
        i_arr = L.tensor_array([L.as_tensor(r) for r in intra])
        t_s, t_t, t_o = L.as_tensor(sample), L.as_tensor(t), L.as_tensor(out)
        t_e = L.as_tensor(ehs) if ehs is not None else None
        t_m = L.as_tensor(mid) if mid is not None else None
        L.check(L.lib().mrisr_unet_forward(self._h, C.byref(t_s), C.byref(t_t), C.byref(t_e) if t_e else None,
                                           d_arr if down else None, len(down), C.byref(t_m) if t_m else None,
                                           i_arr if intra else None, len(intra), C.byref(t_o), L.stream_ptr()))
        return _Output(out) if return_dict else (out,)

    def set_context(self, encoder_hidden_states: torch.Tensor, latent_hw: Tuple[int, int]):
        """Pre-compute the cross-attention K/V of a fixed prompt (then pass encoder_hidden_states=None)."""
        ehs = encoder_hidden_states.to(self.device).contiguous()
        self._ctx_keepalive = ehs
        t = L.as_tensor(ehs)
        L.check(L.lib().mrisr_model_set_context(self._h, C.byref(t), int(latent_hw[0]), int(latent_hw[1]), L.stream_ptr()))


class ControlNetModel(_DeviceModel):
    _create = "mrisr_controlnet_create"

    def __call__(self, sample: torch.Tensor, timestep, encoder_hidden_states: Optional[torch.Tensor] = None,
                 controlnet_cond: Optional[torch.Tensor] = None, conditioning_scale: float = 1.0,
                 return_dict: bool = True, **_unused):
        if not self._finalized:
            raise L.MrisrError("load_state_dict() first")
        sample = sample.to(self.device).contiguous()
        B, _, h, w = sample.shape
        t = self._timestep(timestep, B)
        ehs = encoder_hidden_states.to(self.device).contiguous() if encoder_hidden_states is not None else None
        cond = controlnet_cond.to(self.device).contiguous() if controlnet_cond is not None else None
        if cond is not None and (cond.shape[2] != 8 * h or cond.shape[3] != 8 * w):
            raise ValueError(f"controlnet_cond must be [B,3,{8 * h},{8 * w}]; got {tuple(cond.shape)}")
        shapes = self._skip_shapes(B, h, w)
        outs = [torch.empty(s, dtype=sample.dtype, device=self.device) for s in shapes]
        d_arr = L.tensor_array([L.as_tensor(o) for o in outs[:-1]])
        t_mid = L.as_tensor(outs[-1])
        t_s, t_t = L.as_tensor(sample), L.as_tensor(t)
        t_e = L.as_tensor(ehs) if ehs is not None else None
        t_c = L.as_tensor(cond) if cond is not None else None
        L.check(L.lib().mrisr_controlnet_forward(self._h, C.byref(t_s), C.byref(t_t), C.byref(t_e) if t_e else None,
                                                 C.byref(t_c) if t_c else None, C.c_float(conditioning_scale), d_arr,
                                                 len(outs) - 1, C.byref(t_mid), L.stream_ptr()))
        if return_dict:
            return _ControlNetOutput(outs[:-1], outs[-1])
        return outs[:-1], outs[-1]


class Adapter_XL:
    """T2I-Adapter of the reference (``src/adapters/modules.py:114-157``).  Only ``sk=True`` is runnable in the
    reference; ``sk=False`` raises here too."""

    def __init__(self, channels=(320, 640, 1280, 1280), nums_rb=3, cin=192, ksize=3, sk=True, use_conv=True,
                 compute_dtype="bf16", device="cuda"):
        if not sk:
            raise RuntimeError("Adapter_XL(sk=False) cannot run in the reference either: the skep conv is applied to the "
                               "already-projected tensor (channel mismatch). Use sk=True.")
        if not torch.cuda.is_available():
            raise L.MrisrError("mrisr needs an AMD GPU (gfx950); there is no CPU fallback")
        self.channels, self.nums_rb, self.cin, self.ksize, self.use_conv = tuple(channels), nums_rb, cin, ksize, use_conv
        self.device = torch.device(device)
        c = L.AdapterCfg()
        for i, ch in enumerate(channels):
            c.channels[i] = ch
        c.nums_rb, c.cin, c.ksize, c.use_conv = nums_rb, cin, ksize, 1 if use_conv else 0
        c.compute_dtype = L.dtype_id(compute_dtype)
        self._h = C.c_void_p()
        L.check(L.lib().mrisr_adapter_create(C.byref(c), C.byref(self._h)))
        self._params: Dict[str, torch.Tensor] = {}
        self._finalized = False

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                L.lib().mrisr_adapter_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def eval(self):
        return self

    def state_dict(self):
        return dict(self._params)

    def load_state_dict(self, sd, strict: bool = True):
        fn = L.lib().mrisr_adapter_set_param
        for k, v in sd.items():
            self._params[k] = v
            L.push_param(fn, self._h, k, v)
        L.check(L.lib().mrisr_adapter_finalize(self._h, L.stream_ptr()))
        self._finalized = True
        return self

    def feature_shapes(self, B: int, H: int, W: int):
        h, w = H // 8, W // 8
        return [(B, c, h >> i, w >> i) for i, c in enumerate(self.channels)]

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        if not self._finalized:
            raise L.MrisrError("load_state_dict() first")
        assert x.shape[1] * 64 == self.cin  # same shape guard family as modules.py:71,75
        x = x.to(self.device).contiguous()
        B, _, H, W = x.shape
        feats = [torch.empty(s, dtype=x.dtype, device=self.device) for s in self.feature_shapes(B, H, W)]
        arr = L.tensor_array([L.as_tensor(f) for f in feats])
        t_x = L.as_tensor(x)
        L.check(L.lib().mrisr_adapter_forward(self._h, C.byref(t_x), arr, len(feats), L.stream_ptr()))
        return feats

    __call__ = forward
