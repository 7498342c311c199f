"""Oracle: functional CPU restatement of diffusers' SD-1.5 UNet2DConditionModel,
ControlNetModel and peft LoRA linears.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).  Parity unpinned: diffusers/peft are absent offline; the
spec followed is SURVEY.md Appendix A.1-A.6, anchored on the reference's call
sites ``src/adapters/res_srdiff.py:65-70`` (ControlNet) and ``:73-78`` (UNet).

Everything is a pure function of a flat ``dict[str, Tensor]`` whose keys are the
diffusers state-dict names (App. A.5), so that real SD-1.5 ``.safetensors`` and
the HIP model's ``load_state_dict`` take the very same dictionaries.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class UNetConfig:
    """Mirror of the diffusers config keys the path depends on (App. A.1)."""

    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # True -> CrossAttnDownBlock2D / CrossAttnUpBlock2D at that resolution level
    attn_levels: Tuple[bool, ...] = (True, True, True, False)
    num_heads: int = 8  # diffusers "attention_head_dim": 8 is really the head COUNT
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    # ControlNet only
    cond_channels: int = 3
    cond_embed_channels: Tuple[int, ...] = (16, 32, 96, 256)

    @property
    def time_embed_dim(self) -> int:
        return 4 * self.block_out_channels[0]

    @property
    def num_levels(self) -> int:
        return len(self.block_out_channels)

    def skip_channels(self) -> List[int]:
        """Channels of the 12 (SD-1.5) skip tensors pushed by the down path."""
        ch = [self.block_out_channels[0]]
        for i, c in enumerate(self.block_out_channels):
            ch += [c] * self.layers_per_block
            if i < self.num_levels - 1:
                ch.append(c)
        return ch


SD15 = UNetConfig()
# reduced-width config used for fixtures / fast parity tests.  Channels stay multiples of 64 and
# groups at 32 so the very same HIP kernels (64-wide K tiles, GroupNorm32) run it.
TINY = UNetConfig(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64)
# BASELINE config 1 "MNIST plumbing": 1-channel, two levels, 32x32 (28 padded), CPU-runnable.
MNIST = UNetConfig(in_channels=1, out_channels=1, block_out_channels=(64, 128), attn_levels=(True, False),
                   cross_attention_dim=64)


# --------------------------------------------------------------------------------------
# parameter construction (deterministic, key order = diffusers registration order)
# --------------------------------------------------------------------------------------
class _Init:
    def __init__(self, seed: int, dtype=torch.float32, perturb_norm: bool = False):
        self.g = torch.Generator(device="cpu").manual_seed(seed)
        self.p: Params = {}
        self.dtype = dtype
        self.perturb_norm = perturb_norm

    def _u(self, shape, bound):
        return ((torch.rand(shape, generator=self.g, dtype=torch.float32) * 2 - 1) * bound).to(self.dtype)

    def conv(self, name, cin, cout, k, bias=True):
        bound = 1.0 / math.sqrt(cin * k * k)  # = torch's kaiming_uniform(a=sqrt(5)) bound
        self.p[name + ".weight"] = self._u((cout, cin, k, k), bound)
        if bias:
            self.p[name + ".bias"] = self._u((cout,), bound)

    def linear(self, name, cin, cout, bias=True):
        bound = 1.0 / math.sqrt(cin)
        self.p[name + ".weight"] = self._u((cout, cin), bound)
        if bias:
            self.p[name + ".bias"] = self._u((cout,), bound)

    def norm(self, name, c):
        if self.perturb_norm:
            self.p[name + ".weight"] = (1 + 0.1 * torch.randn(c, generator=self.g)).to(self.dtype)
            self.p[name + ".bias"] = (0.1 * torch.randn(c, generator=self.g)).to(self.dtype)
        else:
            self.p[name + ".weight"] = torch.ones(c, dtype=self.dtype)
            self.p[name + ".bias"] = torch.zeros(c, dtype=self.dtype)

    def resnet(self, name, cin, cout, temb):
        self.norm(name + ".norm1", cin)
        self.conv(name + ".conv1", cin, cout, 3)
        self.linear(name + ".time_emb_proj", temb, cout)
        self.norm(name + ".norm2", cout)
        self.conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            self.conv(name + ".conv_shortcut", cin, cout, 1)

    def transformer(self, name, c, ctx):
        self.norm(name + ".norm", c)
        self.conv(name + ".proj_in", c, c, 1)
        b = name + ".transformer_blocks.0"
        self.norm(b + ".norm1", c)
        for n in ("to_q", "to_k", "to_v"):
            self.linear(f"{b}.attn1.{n}", c, c, bias=False)
        self.linear(b + ".attn1.to_out.0", c, c)
        self.norm(b + ".norm2", c)
        self.linear(b + ".attn2.to_q", c, c, bias=False)
        self.linear(b + ".attn2.to_k", ctx, c, bias=False)
        self.linear(b + ".attn2.to_v", ctx, c, bias=False)
        self.linear(b + ".attn2.to_out.0", c, c)
        self.norm(b + ".norm3", c)
        self.linear(b + ".ff.net.0.proj", c, 8 * c)
        self.linear(b + ".ff.net.2", 4 * c, c)
        self.conv(name + ".proj_out", c, c, 1)

    def encoder(self, cfg: UNetConfig):
        """conv_in, time embedding, down blocks, mid block - shared by UNet and ControlNet."""
        c0 = cfg.block_out_channels[0]
        temb = cfg.time_embed_dim
        self.conv("conv_in", cfg.in_channels, c0, 3)
        self.linear("time_embedding.linear_1", c0, temb)
        self.linear("time_embedding.linear_2", temb, temb)
        cin = c0
        for i, c in enumerate(cfg.block_out_channels):
            for j in range(cfg.layers_per_block):
                self.resnet(f"down_blocks.{i}.resnets.{j}", cin, c, temb)
                cin = c
            if cfg.attn_levels[i]:
                for j in range(cfg.layers_per_block):
                    self.transformer(f"down_blocks.{i}.attentions.{j}", c, cfg.cross_attention_dim)
            if i < cfg.num_levels - 1:
                self.conv(f"down_blocks.{i}.downsamplers.0.conv", c, c, 3)
        cm = cfg.block_out_channels[-1]
        self.resnet("mid_block.resnets.0", cm, cm, temb)
        self.transformer("mid_block.attentions.0", cm, cfg.cross_attention_dim)
        self.resnet("mid_block.resnets.1", cm, cm, temb)


def init_unet_params(cfg: UNetConfig = SD15, seed: int = 20260501, dtype=torch.float32,
                     perturb_norm: bool = False) -> Params:
    it = _Init(seed, dtype, perturb_norm)
    it.encoder(cfg)
    temb = cfg.time_embed_dim
    skips = cfg.skip_channels()
    rev = list(reversed(cfg.block_out_channels))
    prev = rev[0]
    for i, c in enumerate(rev):
        lvl = cfg.num_levels - 1 - i
        for j in range(cfg.layers_per_block + 1):
            it.resnet(f"up_blocks.{i}.resnets.{j}", prev + skips.pop(), c, temb)
            prev = c
        if cfg.attn_levels[lvl]:
            for j in range(cfg.layers_per_block + 1):
                it.transformer(f"up_blocks.{i}.attentions.{j}", c, cfg.cross_attention_dim)
        if i < cfg.num_levels - 1:
            it.conv(f"up_blocks.{i}.upsamplers.0.conv", c, c, 3)
    it.norm("conv_norm_out", cfg.block_out_channels[0])
    it.conv("conv_out", cfg.block_out_channels[0], cfg.out_channels, 3)
    return it.p


def init_controlnet_params(cfg: UNetConfig = SD15, seed: int = 20260502, dtype=torch.float32,
                           perturb_norm: bool = False, zero_init: bool = False) -> Params:
    """diffusers zero-initialises conv_out of the cond embedding and every controlnet_*_block; with
    ``zero_init=False`` (default) they get N(0, 0.02^2) so the paths are exercised."""
    it = _Init(seed, dtype, perturb_norm)
    it.encoder(cfg)
    ce = cfg.cond_embed_channels
    it.conv("controlnet_cond_embedding.conv_in", cfg.cond_channels, ce[0], 3)
    k = 0
    for a, b in zip(ce[:-1], ce[1:]):
        it.conv(f"controlnet_cond_embedding.blocks.{k}", a, a, 3)
        it.conv(f"controlnet_cond_embedding.blocks.{k + 1}", a, b, 3)
        k += 2
    it.conv("controlnet_cond_embedding.conv_out", ce[-1], cfg.block_out_channels[0], 3)
    for k, c in enumerate(cfg.skip_channels()):
        it.conv(f"controlnet_down_blocks.{k}", c, c, 1)
    it.conv("controlnet_mid_block", cfg.block_out_channels[-1], cfg.block_out_channels[-1], 1)
    zeroed = [n for n in it.p if n.startswith(("controlnet_down_blocks", "controlnet_mid_block",
                                                "controlnet_cond_embedding.conv_out"))]
    for n in zeroed:
        if zero_init:
            it.p[n] = torch.zeros_like(it.p[n])
        else:
            it.p[n] = (0.02 * torch.randn(it.p[n].shape, generator=it.g)).to(dtype)
    return it.p


LORA_TARGETS = ("to_q", "to_k", "to_v", "to_out.0")


def lora_target_modules(params: Params) -> List[str]:
    """All attention projections peft would wrap for target_modules=[to_q,to_k,to_v,to_out.0]."""
    out = []
    for k in params:
        if k.endswith(".weight") and ".attn" in k:
            mod = k[: -len(".weight")]
            if mod.endswith(LORA_TARGETS):
                out.append(mod)
    return out


def init_lora_params(params: Params, rank: int = 4, seed: int = 20260503, zero_B: bool = False) -> Params:
    """peft key names: ``<module>.lora_A.default.weight`` [r,in], ``<module>.lora_B.default.weight`` [out,r].
    peft zero-inits B; default here is N(0,0.02^2) so the fused path is exercised."""
    g = torch.Generator().manual_seed(seed)
    out: Params = {}
    for mod in lora_target_modules(params):
        w = params[mod + ".weight"]
        n_out, n_in = w.shape
        bound = 1.0 / math.sqrt(n_in)
        out[mod + ".lora_A.default.weight"] = ((torch.rand((rank, n_in), generator=g) * 2 - 1) * bound).to(w.dtype)
        b = torch.zeros((n_out, rank)) if zero_B else 0.02 * torch.randn((n_out, rank), generator=g)
        out[mod + ".lora_B.default.weight"] = b.to(w.dtype)
    return out


def count_params(p: Params) -> int:
    return sum(v.numel() for v in p.values())


# --------------------------------------------------------------------------------------
# forward pieces
# --------------------------------------------------------------------------------------
def timestep_embedding(t: Tensor, dim: int) -> Tensor:
    """App. A.2: flip_sin_to_cos=True, freq_shift=0 -> [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t.reshape(-1, 1).to(torch.float32) * freqs[None]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def _w(p: Params, name: str) -> Tensor:
    # peft renames a wrapped layer's weight to base_layer.weight; accept both
    k = name + ".weight"
    return p[k] if k in p else p[name + ".base_layer.weight"]


def _b(p: Params, name: str) -> Optional[Tensor]:
    k = name + ".bias"
    if k in p:
        return p[k]
    return p.get(name + ".base_layer.bias")


# BASELINE configs[4] (fp8 projection GEMMs): when set, ``FP8_LINEARS(name, K) -> bool`` selects the linears / 1x1 projections
# that run with fake-quantised OCP e4m3 operands.  "parity unpinned": the reference has no fp8 path (its only reduced-precision
# hook is ``mixed_precision: "fp16"``, nb ResDif c11:38); this restates the scheme the product implements so that its tolerance
# is principled - one scale per weight row (output channel) and per activation row, amax / 448, round-to-nearest-even to
# ``torch.float8_e4m3fn``, exact accumulation.  The LoRA down-projection rows are quantised the same way; the rank-r
# up-projection stays in full precision.
FP8_LINEARS = None


def fp8_fake_quant_rows(t: Tensor) -> Tensor:
    s = t.abs().amax(dim=-1, keepdim=True).clamp_min(1e-20) / 448.0
    return (t / s).to(torch.float8_e4m3fn).to(t.dtype) * s


def linear(p: Params, name: str, x: Tensor, lora_scale: float = 1.0) -> Tensor:
    """y = x W^T + b  (+ (alpha/r) (x A^T) B^T when LoRA tensors for ``name`` are present; a7)."""
    w = _w(p, name)
    ka = name + ".lora_A.default.weight"
    fp8 = FP8_LINEARS is not None and FP8_LINEARS(name, x.shape[-1])
    if fp8:
        x, w = fp8_fake_quant_rows(x), fp8_fake_quant_rows(w)
    y = F.linear(x, w, _b(p, name))
    if ka in p:
        a = fp8_fake_quant_rows(p[ka]) if fp8 else p[ka]
        y = y + lora_scale * F.linear(F.linear(x, a), p[name + ".lora_B.default.weight"])
    return y


def proj1x1(p: Params, name: str, x: Tensor) -> Tensor:
    """proj_in / proj_out of Transformer2DModel: 1x1 convs = a linear over the channel vector of every pixel."""
    if FP8_LINEARS is not None and FP8_LINEARS(name, x.shape[1]):
        B, C, H, W = x.shape
        w = p[name + ".weight"].reshape(-1, C)
        rows = fp8_fake_quant_rows(x.permute(0, 2, 3, 1).reshape(-1, C))
        y = F.linear(rows, fp8_fake_quant_rows(w), p.get(name + ".bias"))
        return y.reshape(B, H, W, -1).permute(0, 3, 1, 2)
    return conv(p, name, x, padding=0)


def conv(p: Params, name: str, x: Tensor, stride: int = 1, padding: int = 1) -> Tensor:
    return F.conv2d(x, p[name + ".weight"], p.get(name + ".bias"), stride=stride, padding=padding)


def group_norm(p: Params, name: str, x: Tensor, groups: int, eps: float) -> Tensor:
    return F.group_norm(x, groups, p[name + ".weight"], p[name + ".bias"], eps)


def layer_norm(p: Params, name: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), p[name + ".weight"], p[name + ".bias"], 1e-5)


def resnet_block(p: Params, name: str, x: Tensor, emb: Tensor, cfg: UNetConfig) -> Tensor:
    """App. A.3."""
    h = conv(p, name + ".conv1", F.silu(group_norm(p, name + ".norm1", x, cfg.norm_num_groups, cfg.norm_eps)))
    h = h + linear(p, name + ".time_emb_proj", F.silu(emb))[:, :, None, None]
    h = conv(p, name + ".conv2", F.silu(group_norm(p, name + ".norm2", h, cfg.norm_num_groups, cfg.norm_eps)))
    if (name + ".conv_shortcut.weight") in p:
        x = conv(p, name + ".conv_shortcut", x, padding=0)
    return x + h


# BASELINE configs[4] (fp8 attention): when True, softmax(q k^T / sqrt(d)) v runs with fake-quantised OCP e4m3 operands and the
# per-head scale rule of the product (csrc/attn.hip, "fp8 attention").  "parity unpinned", like FP8_LINEARS.
FP8_ATTENTION = False


def _e4m3(t: Tensor) -> Tensor:
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(t.dtype)


def sdpa_fp8_fake_quant(q: Tensor, k: Tensor, v: Tensor, bf16_inputs: bool = True) -> Tensor:
    """[B, H, N, d] operands.  Per (batch, head): c = sqrt(amax|q| * sl2 / amax|k|), K8 = e4m3(k c), Q8 = e4m3(q sl2 / c) with
    sl2 = log2(e) / sqrt(d) - their product is the score in log2 units; p = 2^(s - rowmax) is quantised as it is (the product keeps
    p within (0, 2^8], a floating-point format does not care); V8 = e4m3(v / sV), sV = amax|v| / 448; the row sum is the sum of
    the QUANTISED p.  ``bf16_inputs``: q / k / v reach the attention rounded to bf16 (the product's head-major buffers)."""
    d = q.shape[-1]
    if bf16_inputs:
        q, k, v = (t.to(torch.bfloat16).to(t.dtype) for t in (q, k, v))
    sl2 = 1.4426950408889634 * d ** -0.5
    aq = q.abs().amax(dim=(-1, -2), keepdim=True)
    ak = k.abs().amax(dim=(-1, -2), keepdim=True)
    av = v.abs().amax(dim=(-1, -2), keepdim=True)
    c = torch.where((aq > 0) & (ak > 0), (aq * sl2 / ak.clamp_min(1e-30)).sqrt(), torch.ones_like(aq))
    sv = torch.where(av > 0, av / 448.0, torch.ones_like(av))
    q8, k8, v8 = _e4m3(q * (sl2 / c)), _e4m3(k * c), _e4m3(v / sv)
    s = torch.matmul(q8, k8.transpose(-1, -2))
    p8 = _e4m3(torch.exp2(s - s.amax(dim=-1, keepdim=True)))
    return sv * torch.matmul(p8, v8) / p8.sum(dim=-1, keepdim=True)


def attention(p: Params, name: str, x: Tensor, ctx: Tensor, heads: int, lora_scale: float) -> Tensor:
    B, N, C = x.shape
    d = C // heads
    q = linear(p, name + ".to_q", x, lora_scale).view(B, N, heads, d).transpose(1, 2)
    k = linear(p, name + ".to_k", ctx, lora_scale).view(B, -1, heads, d).transpose(1, 2)
    v = linear(p, name + ".to_v", ctx, lora_scale).view(B, -1, heads, d).transpose(1, 2)
    if FP8_ATTENTION:
        o = sdpa_fp8_fake_quant(q, k, v)
    else:
        s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
        o = torch.matmul(torch.softmax(s, dim=-1), v)
    o = o.transpose(1, 2).reshape(B, N, C)
    return linear(p, name + ".to_out.0", o, lora_scale)


def transformer_2d(p: Params, name: str, x: Tensor, ctx: Tensor, cfg: UNetConfig, lora_scale: float) -> Tensor:
    """App. A.4 (use_linear_projection=False, one BasicTransformerBlock, GEGLU feed-forward)."""
    B, C, H, W = x.shape
    res = x
    h = group_norm(p, name + ".norm", x, cfg.norm_num_groups, 1e-6)
    h = proj1x1(p, name + ".proj_in", h)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    b = name + ".transformer_blocks.0"
    n1 = layer_norm(p, b + ".norm1", h)
    h = h + attention(p, b + ".attn1", n1, n1, cfg.num_heads, lora_scale)
    h = h + attention(p, b + ".attn2", layer_norm(p, b + ".norm2", h), ctx, cfg.num_heads, lora_scale)
    ff = linear(p, b + ".ff.net.0.proj", layer_norm(p, b + ".norm3", h))
    u, g = ff.chunk(2, dim=-1)
    h = h + linear(p, b + ".ff.net.2", u * F.gelu(g))
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return proj1x1(p, name + ".proj_out", h) + res


def time_embed(p: Params, t: Tensor, batch: int, cfg: UNetConfig, dtype) -> Tensor:
    t = torch.as_tensor(t)
    if t.ndim == 0:
        t = t[None].expand(batch)
    te = timestep_embedding(t, cfg.block_out_channels[0]).to(dtype)
    return linear(p, "time_embedding.linear_2", F.silu(linear(p, "time_embedding.linear_1", te)))


def _encoder(p: Params, cfg: UNetConfig, x: Tensor, emb: Tensor, ctx: Tensor, lora_scale: float,
             intrablock: Optional[List[Tensor]] = None, last_skip_inplace: bool = True) -> Tuple[Tensor, List[Tensor]]:
    """Down path (no mid block).  ``x`` is already conv_in(sample) (+ ControlNet cond embedding).
    ``last_skip_inplace=False`` is a TEST hook only: the out-of-place reading of the attention-free hand-off (rounds 1-2),
    kept so that tests can show the product follows the in-place one."""
    skips = [x]
    intrablock = list(intrablock) if intrablock is not None else []
    for i in range(cfg.num_levels):
        has_attn = cfg.attn_levels[i]
        for j in range(cfg.layers_per_block):
            x = resnet_block(p, f"down_blocks.{i}.resnets.{j}", x, emb, cfg)
            if has_attn:
                x = transformer_2d(p, f"down_blocks.{i}.attentions.{j}", x, ctx, cfg, lora_scale)
                if j == cfg.layers_per_block - 1 and intrablock:
                    x = x + intrablock.pop(0)  # T2I-Adapter: the skip carries it (App. A.1 step 3)
            skips.append(x)
        if i < cfg.num_levels - 1:
            x = conv(p, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
            skips.append(x)
        if not has_attn and intrablock:
            # attention-free block: diffusers' UNet2DConditionModel.forward does
            #     sample, res_samples = downsample_block(hidden_states=sample, temb=emb)
            #     sample += down_intrablock_additional_residuals.pop(0)
            # and DownBlock2D.forward returns (hidden_states, output_states) with output_states[-1] IS hidden_states
            # (the last resnet's output, or the downsampler's), so the in-place add also lands in res_samples[-1]:
            # the block's LAST skip carries the feature, its earlier skips do not.  Same order as the original
            # TencentARC T2I-Adapter loop (h = h + feature before hs.append(h)).
            x = x + intrablock.pop(0)
            if last_skip_inplace:
                skips[-1] = x
    return x, skips


def _mid(p: Params, cfg: UNetConfig, x: Tensor, emb: Tensor, ctx: Tensor, lora_scale: float) -> Tensor:
    x = resnet_block(p, "mid_block.resnets.0", x, emb, cfg)
    x = transformer_2d(p, "mid_block.attentions.0", x, ctx, cfg, lora_scale)
    return resnet_block(p, "mid_block.resnets.1", x, emb, cfg)


def unet_forward(p: Params, cfg: UNetConfig, sample: Tensor, timestep, encoder_hidden_states: Tensor,
                 down_block_additional_residuals: Optional[Sequence[Tensor]] = None,
                 mid_block_additional_residual: Optional[Tensor] = None,
                 down_intrablock_additional_residuals: Optional[Sequence[Tensor]] = None,
                 lora_scale: float = 1.0, _adapter_last_skip_inplace: bool = True) -> Tensor:
    """eps_hat = UNet(x_t, t, ctx [, ControlNet residuals] [, T2I-Adapter residuals]).  App. A.1.
    (``_adapter_last_skip_inplace``: test hook, see ``_encoder``.)"""
    B = sample.shape[0]
    ctx = encoder_hidden_states
    emb = time_embed(p, timestep, B, cfg, sample.dtype)
    x = conv(p, "conv_in", sample)
    x, skips = _encoder(p, cfg, x, emb, ctx, lora_scale, down_intrablock_additional_residuals, _adapter_last_skip_inplace)
    if down_block_additional_residuals is not None:
        skips = [s + r for s, r in zip(skips, down_block_additional_residuals)]
    x = _mid(p, cfg, x, emb, ctx, lora_scale)
    if mid_block_additional_residual is not None:
        x = x + mid_block_additional_residual
    for i in range(cfg.num_levels):
        lvl = cfg.num_levels - 1 - i
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(p, f"up_blocks.{i}.resnets.{j}", x, emb, cfg)
            if cfg.attn_levels[lvl]:
                x = transformer_2d(p, f"up_blocks.{i}.attentions.{j}", x, ctx, cfg, lora_scale)
        if i < cfg.num_levels - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = conv(p, f"up_blocks.{i}.upsamplers.0.conv", x)
    x = F.silu(group_norm(p, "conv_norm_out", x, cfg.norm_num_groups, cfg.norm_eps))
    return conv(p, "conv_out", x)


def controlnet_cond_embedding(p: Params, cfg: UNetConfig, cond: Tensor) -> Tensor:
    """App. A.6: timestep-independent -> callers may hoist it out of the sampling loop."""
    e = F.silu(conv(p, "controlnet_cond_embedding.conv_in", cond))
    nblk = 2 * (len(cfg.cond_embed_channels) - 1)
    for k in range(nblk):
        e = F.silu(conv(p, f"controlnet_cond_embedding.blocks.{k}", e, stride=2 if k % 2 else 1))
    return conv(p, "controlnet_cond_embedding.conv_out", e)


def controlnet_forward(p: Params, cfg: UNetConfig, sample: Tensor, timestep, encoder_hidden_states: Tensor,
                       controlnet_cond: Tensor, conditioning_scale: float = 1.0,
                       lora_scale: float = 1.0) -> Tuple[List[Tensor], Tensor]:
    """(down_res[12], mid_res) = ControlNet(x_t, t, ctx, cond).  App. A.6."""
    B = sample.shape[0]
    emb = time_embed(p, timestep, B, cfg, sample.dtype)
    x = conv(p, "conv_in", sample) + controlnet_cond_embedding(p, cfg, controlnet_cond)
    x, skips = _encoder(p, cfg, x, emb, encoder_hidden_states, lora_scale)
    x = _mid(p, cfg, x, emb, encoder_hidden_states, lora_scale)
    down = [conv(p, f"controlnet_down_blocks.{k}", s, padding=0) * conditioning_scale for k, s in enumerate(skips)]
    mid = conv(p, "controlnet_mid_block", x, padding=0) * conditioning_scale
    return down, mid


# --------------------------------------------------------------------------------------
# duck-typed objects with the call surface the reference uses
# --------------------------------------------------------------------------------------
class _Out:
    def __init__(self, sample):
        self.sample = sample


class OracleUNet:
    """Callable like diffusers' UNet2DConditionModel at ``res_srdiff.py:73-78``."""

    def __init__(self, params: Params, cfg: UNetConfig = SD15, lora_scale: float = 1.0):
        self.params, self.config, self.lora_scale = params, cfg, lora_scale
        self.calls: List[Tensor] = []  # records every `sample` it was given (trajectory capture)
        self.record = False

    def eval(self):
        return self

    def state_dict(self) -> Params:
        return self.params

    def __call__(self, sample, timestep, encoder_hidden_states=None, down_block_additional_residuals=None,
                 mid_block_additional_residual=None, down_intrablock_additional_residuals=None,
                 return_dict: bool = True):
        if self.record:
            self.calls.append(sample.detach().clone())
        y = unet_forward(self.params, self.config, sample, timestep, encoder_hidden_states,
                         down_block_additional_residuals, mid_block_additional_residual,
                         down_intrablock_additional_residuals, self.lora_scale)
        return _Out(y) if return_dict else (y,)


class OracleControlNet:
    """Callable like diffusers' ControlNetModel at ``res_srdiff.py:65-70``."""

    def __init__(self, params: Params, cfg: UNetConfig = SD15):
        self.params, self.config = params, cfg

    def eval(self):
        return self

    def state_dict(self) -> Params:
        return self.params

    def __call__(self, sample, timestep, encoder_hidden_states=None, controlnet_cond=None,
                 conditioning_scale: float = 1.0, return_dict: bool = True):
        down, mid = controlnet_forward(self.params, self.config, sample, timestep, encoder_hidden_states,
                                       controlnet_cond, conditioning_scale)
        if return_dict:
            class R:  # noqa: N801 - tiny result holder
                down_block_res_samples, mid_block_res_sample = down, mid
            return R
        return down, mid
