"""State-dict templates: every diffusers / peft key of the UNet2DConditionModel / ControlNetModel family
 with its shape, and a device-side random initialiser (PyTorch's default
U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases) for benchmarks - there is no network for checkpoints.
Real SD-1.5 ``.safetensors`` use the same keys and load through ``load_state_dict`` unchanged."""
from __future__ import annotations

import math
from typing import Dict, Iterator, List, Tuple

import torch

from .models import UNetConfig

Shape = Tuple[int, ...]


def _resnet(n: str, cin: int, cout: int, temb: int):
    yield n + ".norm1.weight", (cin,), 0
    yield n + ".norm1.bias", (cin,), -1
    yield n + ".conv1.weight", (cout, cin, 3, 3), cin * 9
    yield n + ".conv1.bias", (cout,), cin * 9
    yield n + ".time_emb_proj.weight", (cout, temb), temb
    yield n + ".time_emb_proj.bias", (cout,), temb
    yield n + ".norm2.weight", (cout,), 0
    yield n + ".norm2.bias", (cout,), -1
    yield n + ".conv2.weight", (cout, cout, 3, 3), cout * 9
    yield n + ".conv2.bias", (cout,), cout * 9
    if cin != cout:
        yield n + ".conv_shortcut.weight", (cout, cin, 1, 1), cin
        yield n + ".conv_shortcut.bias", (cout,), cin


def _transformer(n: str, c: int, ctx: int):
    yield n + ".norm.weight", (c,), 0
    yield n + ".norm.bias", (c,), -1
    yield n + ".proj_in.weight", (c, c, 1, 1), c
    yield n + ".proj_in.bias", (c,), c
    b = n + ".transformer_blocks.0"
    for i, kdim in ((1, c), (2, ctx)):
        yield f"{b}.norm{i}.weight", (c,), 0
        yield f"{b}.norm{i}.bias", (c,), -1
        yield f"{b}.attn{i}.to_q.weight", (c, c), c
        yield f"{b}.attn{i}.to_k.weight", (c, kdim), kdim
        yield f"{b}.attn{i}.to_v.weight", (c, kdim), kdim
        yield f"{b}.attn{i}.to_out.0.weight", (c, c), c
        yield f"{b}.attn{i}.to_out.0.bias", (c,), c
    yield b + ".norm3.weight", (c,), 0
    yield b + ".norm3.bias", (c,), -1
    yield b + ".ff.net.0.proj.weight", (8 * c, c), c
    yield b + ".ff.net.0.proj.bias", (8 * c,), c
    yield b + ".ff.net.2.weight", (c, 4 * c), 4 * c
    yield b + ".ff.net.2.bias", (c,), 4 * c
    yield n + ".proj_out.weight", (c, c, 1, 1), c
    yield n + ".proj_out.bias", (c,), c


def _encoder(cfg: UNetConfig):
    c0, temb, L = cfg.block_out_channels[0], 4 * cfg.block_out_channels[0], len(cfg.block_out_channels)
    yield "conv_in.weight", (c0, cfg.in_channels, 3, 3), cfg.in_channels * 9
    yield "conv_in.bias", (c0,), cfg.in_channels * 9
    yield "time_embedding.linear_1.weight", (temb, c0), c0
    yield "time_embedding.linear_1.bias", (temb,), c0
    yield "time_embedding.linear_2.weight", (temb, temb), temb
    yield "time_embedding.linear_2.bias", (temb,), temb
    cin = c0
    for i, c in enumerate(cfg.block_out_channels):
        attn = cfg.down_block_types[i].startswith("CrossAttn")
        for j in range(cfg.layers_per_block):
            yield from _resnet(f"down_blocks.{i}.resnets.{j}", cin, c, temb)
            cin = c
            if attn:
                yield from _transformer(f"down_blocks.{i}.attentions.{j}", c, cfg.cross_attention_dim)
        if i < L - 1:
            yield f"down_blocks.{i}.downsamplers.0.conv.weight", (c, c, 3, 3), c * 9
            yield f"down_blocks.{i}.downsamplers.0.conv.bias", (c,), c * 9
    cm = cfg.block_out_channels[-1]
    yield from _resnet("mid_block.resnets.0", cm, cm, temb)
    yield from _transformer("mid_block.attentions.0", cm, cfg.cross_attention_dim)
    yield from _resnet("mid_block.resnets.1", cm, cm, temb)


def skip_channels(cfg: UNetConfig) -> List[int]:
    ch = [cfg.block_out_channels[0]]
    for i, c in enumerate(cfg.block_out_channels):
        ch += [c] * cfg.layers_per_block
        if i < len(cfg.block_out_channels) - 1:
            ch.append(c)
    return ch


def unet_param_shapes(cfg: UNetConfig) -> Iterator[Tuple[str, Shape, int]]:
    """(key, shape, fan_in); fan_in 0 = norm weight (ones), -1 = norm bias (zeros)."""
    yield from _encoder(cfg)
    temb, L = 4 * cfg.block_out_channels[0], len(cfg.block_out_channels)
    skips = skip_channels(cfg)
    rev = list(reversed(cfg.block_out_channels))
    prev = rev[0]
    for i, c in enumerate(rev):
        attn = cfg.down_block_types[L - 1 - i].startswith("CrossAttn")
        for j in range(cfg.layers_per_block + 1):
            yield from _resnet(f"up_blocks.{i}.resnets.{j}", prev + skips.pop(), c, temb)
            prev = c
            if attn:
                yield from _transformer(f"up_blocks.{i}.attentions.{j}", c, cfg.cross_attention_dim)
        if i < L - 1:
            yield f"up_blocks.{i}.upsamplers.0.conv.weight", (c, c, 3, 3), c * 9
            yield f"up_blocks.{i}.upsamplers.0.conv.bias", (c,), c * 9
    c0 = cfg.block_out_channels[0]
    yield "conv_norm_out.weight", (c0,), 0
    yield "conv_norm_out.bias", (c0,), -1
    yield "conv_out.weight", (cfg.out_channels, c0, 3, 3), c0 * 9
    yield "conv_out.bias", (cfg.out_channels,), c0 * 9


def controlnet_param_shapes(cfg: UNetConfig) -> Iterator[Tuple[str, Shape, int]]:
    yield from _encoder(cfg)
    ce = cfg.conditioning_embedding_out_channels
    p = "controlnet_cond_embedding"
    yield p + ".conv_in.weight", (ce[0], cfg.conditioning_channels, 3, 3), cfg.conditioning_channels * 9
    yield p + ".conv_in.bias", (ce[0],), cfg.conditioning_channels * 9
    k = 0
    for a, b in zip(ce[:-1], ce[1:]):
        for cin, cout in ((a, a), (a, b)):
            yield f"{p}.blocks.{k}.weight", (cout, cin, 3, 3), cin * 9
            yield f"{p}.blocks.{k}.bias", (cout,), cin * 9
            k += 1
    c0 = cfg.block_out_channels[0]
    yield p + ".conv_out.weight", (c0, ce[-1], 3, 3), ce[-1] * 9
    yield p + ".conv_out.bias", (c0,), ce[-1] * 9
    for k, c in enumerate(skip_channels(cfg)):
        yield f"controlnet_down_blocks.{k}.weight", (c, c, 1, 1), c
        yield f"controlnet_down_blocks.{k}.bias", (c,), c
    cm = cfg.block_out_channels[-1]
    yield "controlnet_mid_block.weight", (cm, cm, 1, 1), cm
    yield "controlnet_mid_block.bias", (cm,), cm


LORA_TARGETS = ("to_q", "to_k", "to_v", "to_out.0")


def lora_param_shapes(cfg: UNetConfig, rank: int) -> Iterator[Tuple[str, Shape, int]]:
    """peft keys for target_modules = to_q,to_k,to_v,to_out.0 of every attention (config keys utils.py:69-70)."""
    for key, shape, _ in unet_param_shapes(cfg):
        if ".attn" in key and key.endswith(".weight"):
            mod = key[: -len(".weight")]
            if mod.endswith(LORA_TARGETS):
                yield mod + ".lora_A.default.weight", (rank, shape[1]), shape[1]
                yield mod + ".lora_B.default.weight", (shape[0], rank), -2


def random_state_dict(shapes, seed: int, device="cuda", lora_B_std: float = 0.02) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for key, shape, fan in shapes:
        if fan == 0:
            out[key] = torch.ones(shape, device=device)
        elif fan == -1:
            out[key] = torch.zeros(shape, device=device)
        elif fan == -2:  # peft zero-inits lora_B; a small normal keeps the fused path exercised
            out[key] = torch.randn(shape, device=device, generator=g) * lora_B_std
        else:
            bound = 1.0 / math.sqrt(fan)
            out[key] = (torch.rand(shape, device=device, generator=g) * 2 - 1) * bound
    return out
