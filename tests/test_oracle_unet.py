"""CPU: known-answer checks of the UNet / ControlNet / LoRA oracle (parity unpinned):
parameter counts, diffusers key/shape list, op cross-checks against torch.nn.functional."""
import math

import pytest
import torch

from oracle import schedulers as osch
from oracle import unet as ou

torch.set_grad_enabled(False)




def _resnet(cin, cout, temb):
    n = 2 * cin + (cin * cout * 9 + cout) + (temb * cout + cout) + 2 * cout + (cout * cout * 9 + cout)
    if cin != cout:
        n += cin * cout + cout
    return n


def _transformer(c, ctx):
    n = 2 * c + 2 * (c * c + c)  # norm, proj_in, proj_out
    n += 3 * 2 * c  # 3 layer norms
    n += 3 * c * c + (c * c + c)  # attn1
    n += c * c + 2 * ctx * c + (c * c + c)  # attn2
    n += (c * 8 * c + 8 * c) + (4 * c * c + c)  # GEGLU ff
    return n


def test_sd15_param_counts_by_formula():
    cfg = ou.SD15
    temb = cfg.time_embed_dim
    n = 4 * 320 * 9 + 320 + (320 * temb + temb) + (temb * temb + temb)
    cin = 320
    for i, c in enumerate(cfg.block_out_channels):
        for j in range(2):
            n += _resnet(cin, c, temb)
            cin = c
            if cfg.attn_levels[i]:
                n += _transformer(c, 768)
        if i < 3:
            n += c * c * 9 + c
    enc_mid = n + 2 * _resnet(1280, 1280, temb) + _transformer(1280, 768)
    n = enc_mid
    skips = cfg.skip_channels()
    assert skips == [320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280]
    prev = 1280
    for i, c in enumerate(reversed(cfg.block_out_channels)):
        for j in range(3):
            n += _resnet(prev + skips.pop(), c, temb)
            prev = c
            if cfg.attn_levels[3 - i]:
                n += _transformer(c, 768)
        if i < 3:
            n += c * c * 9 + c
    n += 2 * 320 + 320 * 4 * 9 + 4
    assert n == 859_520_964
    # ControlNet = encoder + mid + cond embedding + zero convs
    ce = cfg.cond_embed_channels
    m = enc_mid + 3 * ce[0] * 9 + ce[0]
    for a, b in zip(ce[:-1], ce[1:]):
        m += a * a * 9 + a + a * b * 9 + b
    m += ce[-1] * 320 * 9 + 320
    m += sum(c * c + c for c in cfg.skip_channels()) + 1280 * 1280 + 1280
    assert m == 361_279_120
    # LoRA r=4 on to_q/k/v/out of all 32 attentions: r*(14C+1536) per transformer block
    lora = sum(4 * (14 * c + 2 * 768) for c in [320] * 5 + [640] * 5 + [1280] * 6)
    assert lora == 797_184


@pytest.mark.parametrize("cfg", [ou.TINY, ou.MNIST])
def test_small_config_keys_and_forward(cfg):
    p = ou.init_unet_params(cfg, seed=1)
    # diffusers key families present (App. A.5)
    for k in ("conv_in.weight", "time_embedding.linear_1.weight", "down_blocks.0.resnets.0.norm1.weight",
              "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight",
              "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
              "down_blocks.0.downsamplers.0.conv.weight", "mid_block.attentions.0.proj_out.bias",
              "up_blocks.0.resnets.0.conv_shortcut.weight", "conv_norm_out.weight", "conv_out.bias"):
        assert k in p, k
    assert "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.bias" not in p
    B, h = 2, 16
    x = torch.randn(B, cfg.in_channels, h, h)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim)
    y = ou.unet_forward(p, cfg, x, torch.tensor(500), ctx)
    assert y.shape == (B, cfg.out_channels, h, h) and torch.isfinite(y).all()
    # scalar t == broadcast [B] t
    y2 = ou.unet_forward(p, cfg, x, torch.tensor([500, 500]), ctx)
    assert torch.allclose(y, y2, atol=1e-6)


def test_lora_keys_and_zero_B_is_identity():
    cfg = ou.TINY
    p = ou.init_unet_params(cfg, seed=2)
    lora0 = ou.init_lora_params(p, rank=4, zero_B=True)
    lora = ou.init_lora_params(p, rank=4)
    n_attn = sum(1 for k in p if k.endswith("attn1.to_q.weight")) * 2
    assert len(lora) == n_attn * 4 * 2
    assert ou.count_params(lora) == sum(
        4 * (p[m + ".weight"].shape[0] + p[m + ".weight"].shape[1]) for m in ou.lora_target_modules(p))
    x = torch.randn(1, 4, 8, 8)
    ctx = torch.randn(1, 77, cfg.cross_attention_dim)
    y = ou.unet_forward(p, cfg, x, torch.tensor(10), ctx)
    y0 = ou.unet_forward({**p, **lora0}, cfg, x, torch.tensor(10), ctx)
    y1 = ou.unet_forward({**p, **lora}, cfg, x, torch.tensor(10), ctx)
    assert torch.equal(y, y0)
    assert not torch.allclose(y, y1, atol=1e-5)
    # merged weights W + s*B@A give the same function (what a fused GEMM tail computes)
    merged = dict(p)
    for m in ou.lora_target_modules(p):
        merged[m + ".weight"] = p[m + ".weight"] + lora[m + ".lora_B.default.weight"] @ lora[m + ".lora_A.default.weight"]
    ym = ou.unet_forward(merged, cfg, x, torch.tensor(10), ctx)
    assert torch.allclose(ym, y1, atol=2e-5)


def test_controlnet_shapes_and_residual_injection():
    cfg = ou.TINY
    up = ou.init_unet_params(cfg, seed=3)
    cp = ou.init_controlnet_params(cfg, seed=4)
    x = torch.randn(1, 4, 8, 8)
    ctx = torch.randn(1, 77, cfg.cross_attention_dim)
    cond = torch.randn(1, 3, 64, 64)
    down, mid = ou.controlnet_forward(cp, cfg, x, torch.tensor(7), ctx, cond)
    assert [d.shape[1] for d in down] == cfg.skip_channels()
    assert [d.shape[-1] for d in down] == [8, 8, 8, 4, 4, 4, 2, 2, 2, 1, 1, 1]
    assert mid.shape == (1, 256, 1, 1)
    y0 = ou.unet_forward(up, cfg, x, torch.tensor(7), ctx)
    y1 = ou.unet_forward(up, cfg, x, torch.tensor(7), ctx, down, mid)
    assert not torch.allclose(y0, y1)
    zc = ou.init_controlnet_params(cfg, seed=4, zero_init=True)
    down0, mid0 = ou.controlnet_forward(zc, cfg, x, torch.tensor(7), ctx, cond)
    assert all(float(d.abs().max()) == 0 for d in down0) and float(mid0.abs().max()) == 0


def test_adapter_feature_of_the_attention_free_block_lands_in_its_last_skip():
    """diffusers: `sample += down_intrablock_additional_residuals.pop(0)` is IN PLACE on the tensor DownBlock2D also returned
    as res_samples[-1] -> the last of the 12 skips carries feature 3, skip 10 (the block's first resnet) does not; features
    0-2 (cross-attention blocks) land in the skip of the block's last (resnet, attention) pair, not in the downsampler's."""
    cfg = ou.TINY
    up = ou.init_unet_params(cfg, seed=3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, cfg.block_out_channels[0], 8, 8), generator=g)
    ctx = torch.randn((1, 77, cfg.cross_attention_dim), generator=g)
    emb = ou.time_embed(up, torch.tensor(7), 1, cfg, x.dtype)
    feats = [torch.randn((1, c, 8 >> i, 8 >> i), generator=g) for i, c in enumerate(cfg.block_out_channels)]
    feats[3] = torch.randn((1, cfg.block_out_channels[3], 1, 1), generator=g)  # level 3 runs at the level-2 downsampler's size
    x0, s0 = ou._encoder(up, cfg, x, emb, ctx, 1.0)
    x1, s1 = ou._encoder(up, cfg, x, emb, ctx, 1.0, [f.clone() for f in feats])
    assert len(s0) == len(s1) == 12
    # level 0: skips 0 (conv_in), 1, 2 (resnet+attn pairs), 3 (downsampler)
    assert torch.equal(s1[0], s0[0]) and torch.equal(s1[1], s0[1])
    assert torch.allclose(s1[2], s0[2] + feats[0], atol=1e-6)
    # a feature of level 3 alone: only the LAST skip and the mid input move, by exactly the feature
    x3, s3 = ou._encoder(up, cfg, x, emb, ctx, 1.0, None)
    only3 = [torch.zeros_like(f) for f in feats[:3]] + [feats[3]]
    x4, s4 = ou._encoder(up, cfg, x, emb, ctx, 1.0, only3)
    for k in range(11):
        assert torch.equal(s4[k], s3[k]), k
    assert torch.allclose(s4[11], s3[11] + feats[3], atol=1e-6) and torch.equal(x4, s4[11])
    # the out-of-place reading (test hook) leaves skip 11 alone - that is what rounds 1-2 computed
    x5, s5 = ou._encoder(up, cfg, x, emb, ctx, 1.0, only3, last_skip_inplace=False)
    assert torch.equal(s5[11], s3[11]) and torch.equal(x5, x4)


def test_time_embedding_layout():
    e = ou.timestep_embedding(torch.tensor([0, 1, 999]), 320)
    assert e.shape == (3, 320)
    assert torch.allclose(e[0, :160], torch.ones(160)) and torch.allclose(e[0, 160:], torch.zeros(160))
    f1 = math.exp(-math.log(10000.0) * 1 / 160)
    assert e[1, 1].item() == pytest.approx(math.cos(f1), rel=1e-6)
    assert e[1, 161].item() == pytest.approx(math.sin(f1), rel=1e-6)


def test_scheduler_tables_and_ddim():
    s = osch.OracleScheduler(timestep_spacing="leading", steps_offset=1)
    ac = s.alphas_cumprod
    assert ac.shape == (1000,) and ac[0].item() == pytest.approx(1 - 0.00085, rel=1e-6)
    assert ac[999].item() == pytest.approx(0.0046600, rel=2e-3)  # SD-1.5 terminal alpha-bar
    s.set_timesteps(50)
    assert s.timesteps[:3].tolist() == [981, 961, 941] and s.timesteps[-1].item() == 1
    tr = osch.make_timesteps(20, spacing="trailing")
    assert tr[0] == 999 and tr[-1] == 49 and len(tr) == 20
    # DDIM with a perfect eps recovers x0 at the last step's alpha_prev = alpha[0]
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(1, 4, 8, 8, generator=g)
    eps = torch.randn(1, 4, 8, 8, generator=g)
    t = 981
    x = ac[t].sqrt() * x0 + (1 - ac[t]).sqrt() * eps
    xp = s.ddim_step(eps, t, x)
    tp = t - 20
    assert torch.allclose(xp, ac[tp].sqrt() * x0 + (1 - ac[tp]).sqrt() * eps, atol=1e-5)
    cx, ce = s.ddim_coeffs(t)
    assert torch.allclose(xp, cx * x + ce * eps, atol=1e-5)


@pytest.mark.parametrize("heads,d,nq,nk", [(8, 40, 64, 64), (8, 80, 16, 77), (8, 160, 16, 16), (2, 8, 5, 3)])
def test_attention_matches_scaled_dot_product_attention(heads, d, nq, nk):
    """the oracle's hand-rolled softmax(q k^T / sqrt(d)) v (oracle/unet.py `attention`, which restates diffusers'
    AttnProcessor) against torch's own F.scaled_dot_product_attention on the same projections - self-attention and the
    77-token cross-attention shape, the three SD-1.5 head sizes, with and without LoRA on the projections."""
    import torch.nn.functional as F
    C = heads * d
    g = torch.Generator().manual_seed(7 + d)
    p = {}
    for n, cin in (("to_q", C), ("to_k", C if nk == nq else 96), ("to_v", C if nk == nq else 96)):
        p[f"a.{n}.weight"] = torch.randn((C, cin), generator=g) / math.sqrt(cin)
    p["a.to_out.0.weight"] = torch.randn((C, C), generator=g) / math.sqrt(C)
    p["a.to_out.0.bias"] = torch.randn((C,), generator=g)
    x = torch.randn((2, nq, C), generator=g)
    ctx = x if nk == nq else torch.randn((2, nk, 96), generator=g)
    for with_lora in (False, True):
        if with_lora:
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                cin = p[f"a.{n}.weight"].shape[1]
                p[f"a.{n}.lora_A.default.weight"] = torch.randn((4, cin), generator=g) / math.sqrt(cin)
                p[f"a.{n}.lora_B.default.weight"] = 0.1 * torch.randn((C, 4), generator=g)
        got = ou.attention(p, "a", x, ctx, heads, 0.5)

        def lin(n, t):
            y = F.linear(t, p[f"a.{n}.weight"], p.get(f"a.{n}.bias"))
            if with_lora:  # peft: y += scale * B(A(x))
                y = y + 0.5 * F.linear(F.linear(t, p[f"a.{n}.lora_A.default.weight"]), p[f"a.{n}.lora_B.default.weight"])
            return y

        q = lin("to_q", x).view(2, nq, heads, d).transpose(1, 2)
        k = lin("to_k", ctx).view(2, nk, heads, d).transpose(1, 2)
        v = lin("to_v", ctx).view(2, nk, heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
        want = lin("to_out.0", o.transpose(1, 2).reshape(2, nq, C))
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), float((got - want).abs().max())


def test_fp8_fake_quant_is_e4m3_round_to_nearest_with_row_scales():
    """The oracle's fake-quant (BASELINE configs[4]): exact on e4m3-representable rows, max at 448, <= 2^-4 relative error
    elsewhere (3 mantissa bits), scale per row."""
    grid = torch.tensor([[0.0, 1.0, 1.125, 1.25, 1.5, 1.75, 2.0, 3.5, 448.0, -448.0, 0.015625, 208.0]])
    assert torch.equal(ou.fp8_fake_quant_rows(grid), grid)                      # amax = 448 -> scale 1: representable values survive
    x = torch.randn(5, 320, generator=torch.Generator().manual_seed(3)) * torch.tensor([[1e-3], [1.0], [50.0], [1e4], [7.0]])
    q = ou.fp8_fake_quant_rows(x)
    s = x.abs().amax(-1, keepdim=True) / 448
    assert torch.allclose(q.abs().amax(-1), x.abs().amax(-1))                   # the row maximum maps to 448 exactly
    big = x.abs() > 16 * 2.0 ** -6 * s                                           # normal range of e4m3 after scaling
    assert float(((q - x).abs() / x.abs())[big].max()) <= 2.0 ** -4 + 1e-6
    # the hook: only the selected linears change, and by a bounded amount
    p = {"l.weight": torch.randn(64, 320), "l.bias": torch.randn(64)}
    xin = torch.randn(3, 7, 320)
    ref = ou.linear(p, "l", xin)
    try:
        ou.FP8_LINEARS = lambda name, K: K == 320
        got = ou.linear(p, "l", xin)
        ou.FP8_LINEARS = lambda name, K: K == 640
        assert torch.equal(ou.linear(p, "l", xin), ref)
    finally:
        ou.FP8_LINEARS = None
    e = float((got - ref).norm() / ref.norm())
    assert 1e-3 < e < 6e-2, e
