"""``AutoencoderKL`` - host mirror of the diffusers VAE surface the reference touches (SURVEY.md 8b "VAE object"):

    vae.encode(x).latent_dist.sample() * vae.config.scaling_factor          src/adapters/res_srdiff.py:49-50
    vae.decode(latents / vae.config.scaling_factor).sample                  src/adapters/res_srdiff.py:107-110

The convolutions / norms / attention run in ``libmrisr.so`` (``csrc/vae.hip``); drawing the posterior noise stays on the
host side so that the global torch RNG stream is consumed exactly as the reference consumes it.  No CPU fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Iterator, Optional, Tuple

import torch

from . import _lib as L


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215

    @classmethod
    def from_oracle_like(cls, cfg) -> "VAEConfig":
        return cls(**{k: getattr(cfg, k) for k in ("in_channels", "out_channels", "latent_channels", "block_out_channels",
                                                   "layers_per_block", "norm_num_groups", "scaling_factor")})


class DiagonalGaussianDistribution:
    """diffusers' posterior object: ``.sample(generator=None)``, ``.mode()``, ``.mean``, ``.logvar``, ``.std``."""

    def __init__(self, moments: torch.Tensor):
        self.parameters = moments
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class _EncoderOutput:
    def __init__(self, dist):
        self.latent_dist = dist


class _DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKL:
    def __init__(self, config=None, compute_dtype="bf16", device="cuda"):
        if not torch.cuda.is_available():
            raise L.MrisrError("mrisr needs an AMD GPU (gfx950); there is no CPU fallback")
        cfg = config if isinstance(config, VAEConfig) else (VAEConfig() if config is None else VAEConfig.from_oracle_like(config))
        self.config = cfg
        self.device = torch.device(device)
        self.compute_dtype = L.torch_dtype(L.dtype_id(compute_dtype))
        c = L.VaeCfg()
        c.in_channels, c.out_channels, c.latent_channels = cfg.in_channels, cfg.out_channels, cfg.latent_channels
        c.num_levels = len(cfg.block_out_channels)
        for i, ch in enumerate(cfg.block_out_channels):
            c.block_out_channels[i] = ch
        c.layers_per_block, c.norm_num_groups = cfg.layers_per_block, cfg.norm_num_groups
        c.compute_dtype, c.scaling_factor = L.dtype_id(compute_dtype), cfg.scaling_factor
        self._h = C.c_void_p()
        L.check(L.lib().mrisr_vae_create(C.byref(c), C.byref(self._h)))
        self._params: Dict[str, torch.Tensor] = {}
        self._finalized = False

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                L.lib().mrisr_vae_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def requires_grad_(self, flag: bool = True):
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self._params)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """diffusers AutoencoderKL key names (encoder.*, decoder.*, quant_conv.*, post_quant_conv.*)."""
        fn = L.lib().mrisr_vae_set_param
        for k, v in sd.items():
            self._params[k] = v
            L.push_param(fn, self._h, k, v)
        L.check(L.lib().mrisr_vae_finalize(self._h, L.stream_ptr()))  # raises on missing keys
        self._finalized = True
        return self

    @property
    def num_parameters(self) -> int:
        return int(L.lib().mrisr_vae_num_params(self._h))

    def _check(self, x: torch.Tensor, ch: int, what: str) -> torch.Tensor:
        if not self._finalized:
            raise L.MrisrError("load_state_dict() first")
        if x.ndim != 4 or x.shape[1] != ch:
            raise ValueError(f"{what} must be [B,{ch},H,W]; got {tuple(x.shape)}")
        return x.to(self.device).contiguous()

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        x = self._check(x, self.config.in_channels, "image")
        f = 2 ** (len(self.config.block_out_channels) - 1)
        if x.shape[2] % f or x.shape[3] % f:
            raise ValueError(f"image height/width must be divisible by {f}; got {tuple(x.shape)}")
        mom = torch.empty((x.shape[0], 2 * self.config.latent_channels, x.shape[2] // f, x.shape[3] // f), dtype=x.dtype,
                          device=self.device)
        t_x, t_m = L.as_tensor(x), L.as_tensor(mom)
        L.check(L.lib().mrisr_vae_encode(self._h, C.byref(t_x), C.byref(t_m), L.stream_ptr()))
        dist = DiagonalGaussianDistribution(mom)
        return _EncoderOutput(dist) if return_dict else (dist,)

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        z = self._check(z, self.config.latent_channels, "latents")
        f = 2 ** (len(self.config.block_out_channels) - 1)
        out = torch.empty((z.shape[0], self.config.out_channels, z.shape[2] * f, z.shape[3] * f), dtype=z.dtype, device=self.device)
        t_z, t_o = L.as_tensor(z), L.as_tensor(out)
        L.check(L.lib().mrisr_vae_decode(self._h, C.byref(t_z), C.byref(t_o), L.stream_ptr()))
        return _DecoderOutput(out) if return_dict else (out,)


def vae_param_shapes(cfg: VAEConfig) -> Iterator[Tuple[str, Tuple[int, ...], int]]:
    """(key, shape, fan_in) of every AutoencoderKL parameter, for ``params.random_state_dict`` (fan_in 0 / -1: norm
    weight / bias)."""
    ch = cfg.block_out_channels
    nl = len(ch)

    def conv(n, cin, cout, k):
        yield n + ".weight", (cout, cin, k, k), cin * k * k
        yield n + ".bias", (cout,), cin * k * k

    def norm(n, c):
        yield n + ".weight", (c,), 0
        yield n + ".bias", (c,), -1

    def resnet(n, cin, cout):
        yield from norm(n + ".norm1", cin)
        yield from conv(n + ".conv1", cin, cout, 3)
        yield from norm(n + ".norm2", cout)
        yield from conv(n + ".conv2", cout, cout, 3)
        if cin != cout:
            yield from conv(n + ".conv_shortcut", cin, cout, 1)

    def mid(n, c):
        yield from resnet(n + ".resnets.0", c, c)
        a = n + ".attentions.0"
        yield from norm(a + ".group_norm", c)
        for m in ("to_q", "to_k", "to_v", "to_out.0"):
            yield f"{a}.{m}.weight", (c, c), c
            yield f"{a}.{m}.bias", (c,), c
        yield from resnet(n + ".resnets.1", c, c)

    yield from conv("encoder.conv_in", cfg.in_channels, ch[0], 3)
    cin = ch[0]
    for i in range(nl):
        for j in range(cfg.layers_per_block):
            yield from resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin, ch[i])
            cin = ch[i]
        if i < nl - 1:
            yield from conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", ch[i], ch[i], 3)
    yield from mid("encoder.mid_block", ch[-1])
    yield from norm("encoder.conv_norm_out", ch[-1])
    yield from conv("encoder.conv_out", ch[-1], 2 * cfg.latent_channels, 3)
    yield from conv("decoder.conv_in", cfg.latent_channels, ch[-1], 3)
    yield from mid("decoder.mid_block", ch[-1])
    rev = list(reversed(ch))
    cin = rev[0]
    for i in range(nl):
        for j in range(cfg.layers_per_block + 1):
            yield from resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin, rev[i])
            cin = rev[i]
        if i < nl - 1:
            yield from conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", rev[i], rev[i], 3)
    yield from norm("decoder.conv_norm_out", rev[-1])
    yield from conv("decoder.conv_out", rev[-1], cfg.out_channels, 3)
    yield from conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    yield from conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
