"""ORACLE (test infrastructure only - never imported by the product): CPU restatement of the SD-1.5 ``AutoencoderKL``
that sits either side of the reference's sampling loop:

    latents = vae.encode(lr).latent_dist.sample() * vae.config.scaling_factor      src/adapters/res_srdiff.py:49-50
    image   = vae.decode(latents / vae.config.scaling_factor).sample               src/adapters/res_srdiff.py:107-110

The arithmetic lives in the un-vendored third-party ``diffusers`` (AutoencoderKL); it is restated
here from the published SD-1.5 VAE architecture under diffusers' state-dict key names.  Parity is pinned by the
known-answer parameter count 83,653,863 of ``stable-diffusion-v1-5/vae`` (tests/test_oracle_vae.py); there are no golden
vectors for it in the reference ("parity unpinned" beyond that count).

config: in/out 3 channels, latent 4, block_out_channels (128, 256, 512, 512), layers_per_block 2, norm_num_groups 32
(eps 1e-6), SiLU, mid-block single-head attention (dim 512), scaling_factor 0.18215.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215


SD15_VAE = VAEConfig()
TINY_VAE = VAEConfig(block_out_channels=(64, 64, 128, 128))


class _Init:
    def __init__(self, seed: int):
        self.g = torch.Generator(device="cpu").manual_seed(seed)
        self.p: Params = {}

    def conv(self, name, cin, cout, k):
        bound = 1.0 / math.sqrt(cin * k * k)
        self.p[name + ".weight"] = (torch.rand((cout, cin, k, k), generator=self.g) * 2 - 1) * bound
        self.p[name + ".bias"] = (torch.rand((cout,), generator=self.g) * 2 - 1) * bound

    def linear(self, name, cin, cout):
        bound = 1.0 / math.sqrt(cin)
        self.p[name + ".weight"] = (torch.rand((cout, cin), generator=self.g) * 2 - 1) * bound
        self.p[name + ".bias"] = (torch.rand((cout,), generator=self.g) * 2 - 1) * bound

    def norm(self, name, c):
        # perturbed affine parameters so that gamma / beta are exercised
        self.p[name + ".weight"] = 1.0 + 0.1 * torch.randn((c,), generator=self.g)
        self.p[name + ".bias"] = 0.1 * torch.randn((c,), generator=self.g)

    def resnet(self, name, cin, cout):
        self.norm(name + ".norm1", cin)
        self.conv(name + ".conv1", cin, cout, 3)
        self.norm(name + ".norm2", cout)
        self.conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            self.conv(name + ".conv_shortcut", cin, cout, 1)

    def mid(self, name, c):
        self.resnet(name + ".resnets.0", c, c)
        a = name + ".attentions.0"
        self.norm(a + ".group_norm", c)
        for m in ("to_q", "to_k", "to_v", "to_out.0"):
            self.linear(a + "." + m, c, c)
        self.resnet(name + ".resnets.1", c, c)


def init_vae_params(cfg: VAEConfig = SD15_VAE, seed: int = 20260505) -> Params:
    """Random-init parameters under diffusers' AutoencoderKL key names, in registration order."""
    it = _Init(seed)
    ch = cfg.block_out_channels
    L = len(ch)
    # encoder
    it.conv("encoder.conv_in", cfg.in_channels, ch[0], 3)
    cin = ch[0]
    for i in range(L):
        for j in range(cfg.layers_per_block):
            it.resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin, ch[i])
            cin = ch[i]
        if i < L - 1:
            it.conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", ch[i], ch[i], 3)
    it.mid("encoder.mid_block", ch[-1])
    it.norm("encoder.conv_norm_out", ch[-1])
    it.conv("encoder.conv_out", ch[-1], 2 * cfg.latent_channels, 3)
    # decoder
    it.conv("decoder.conv_in", cfg.latent_channels, ch[-1], 3)
    it.mid("decoder.mid_block", ch[-1])
    rev = list(reversed(ch))
    cin = rev[0]
    for i in range(L):
        for j in range(cfg.layers_per_block + 1):
            it.resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin, rev[i])
            cin = rev[i]
        if i < L - 1:
            it.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", rev[i], rev[i], 3)
    it.norm("decoder.conv_norm_out", rev[-1])
    it.conv("decoder.conv_out", rev[-1], cfg.out_channels, 3)
    it.conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    it.conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    return it.p


def count_params(p: Params) -> int:
    return sum(v.numel() for v in p.values())


def _conv(p, name, x, stride=1, padding=1):
    return F.conv2d(x, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=padding)


def _gn(p, name, x, cfg):
    return F.group_norm(x, cfg.norm_num_groups, p[name + ".weight"], p[name + ".bias"], cfg.norm_eps)


def _resnet(p, name, x, cfg):
    h = _conv(p, name + ".conv1", F.silu(_gn(p, name + ".norm1", x, cfg)))
    h = _conv(p, name + ".conv2", F.silu(_gn(p, name + ".norm2", h, cfg)))
    if name + ".conv_shortcut.weight" in p:
        x = _conv(p, name + ".conv_shortcut", x, padding=0)
    return x + h


def _attention(p, name, x, cfg):
    """diffusers ``Attention(heads=1, residual_connection=True, norm_num_groups=32)`` on the flattened feature map."""
    B, C, H, W = x.shape
    t = _gn(p, name + ".group_norm", x, cfg).view(B, C, H * W).transpose(1, 2)
    q = F.linear(t, p[name + ".to_q.weight"], p[name + ".to_q.bias"])
    k = F.linear(t, p[name + ".to_k.weight"], p[name + ".to_k.bias"])
    v = F.linear(t, p[name + ".to_v.weight"], p[name + ".to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1) @ v
    o = F.linear(a, p[name + ".to_out.0.weight"], p[name + ".to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def _mid(p, name, x, cfg):
    x = _resnet(p, name + ".resnets.0", x, cfg)
    x = _attention(p, name + ".attentions.0", x, cfg)
    return _resnet(p, name + ".resnets.1", x, cfg)


def encode_moments(p: Params, cfg: VAEConfig, x: Tensor) -> Tensor:
    """[B,3,H,W] in [-1,1] -> [B, 2*latent, H/8, W/8] = (mean | logvar) of the diagonal Gaussian posterior."""
    L = len(cfg.block_out_channels)
    h = _conv(p, "encoder.conv_in", x)
    for i in range(L):
        for j in range(cfg.layers_per_block):
            h = _resnet(p, f"encoder.down_blocks.{i}.resnets.{j}", h, cfg)
        if i < L - 1:
            # Downsample2D(padding=0): asymmetric zero pad (right, bottom), then 3x3 stride 2
            h = _conv(p, f"encoder.down_blocks.{i}.downsamplers.0.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = _mid(p, "encoder.mid_block", h, cfg)
    h = _conv(p, "encoder.conv_out", F.silu(_gn(p, "encoder.conv_norm_out", h, cfg)))
    return _conv(p, "quant_conv", h, padding=0)


def sample_latents(moments: Tensor, noise: Tensor) -> Tensor:
    """DiagonalGaussianDistribution.sample(): mean + exp(0.5 * clamp(logvar, -30, 20)) * noise."""
    mean, logvar = moments.chunk(2, dim=1)
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise


def decode(p: Params, cfg: VAEConfig, z: Tensor) -> Tensor:
    """[B,latent,h,w] (already divided by scaling_factor) -> [B,3,8h,8w]."""
    L = len(cfg.block_out_channels)
    h = _conv(p, "post_quant_conv", z, padding=0)
    h = _conv(p, "decoder.conv_in", h)
    h = _mid(p, "decoder.mid_block", h, cfg)
    for i in range(L):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(p, f"decoder.up_blocks.{i}.resnets.{j}", h, cfg)
        if i < L - 1:
            h = _conv(p, f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    return _conv(p, "decoder.conv_out", F.silu(_gn(p, "decoder.conv_norm_out", h, cfg)))
