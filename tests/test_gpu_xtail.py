"""The fused row-local middle of the C = 320 transformer blocks (csrc/xtail.hip: attn1.to_out + residual -> LayerNorm2 -> attn2.to_q ->
cross-attention over the cached prompt K / V -> attn2.to_out + residual in ONE launch) against (a) the four separate launches it
replaces and (b) the CPU oracle (oracle/unet.py, diffusers' BasicTransformerBlock restated; call site res_srdiff.py:73-78)."""
import ctypes as C
import json

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _cfg():
    from oracle import unet as ou
    # level 0: C = 320 with attention (8 heads of 40 channels: the geometry of the kernel); level 1 and the mid block: C = 640
    return ou.UNetConfig(block_out_channels=(320, 640), attn_levels=(True, False), layers_per_block=1, cross_attention_dim=64)


def _classes(lib, fn):
    fn()
    lib.mrisr_prof_reset(); lib.mrisr_prof_enable(1)
    out = fn().clone()
    torch.cuda.synchronize(); lib.mrisr_prof_enable(0)
    buf = C.create_string_buffer(1 << 20)
    n = lib.mrisr_prof_report(buf, len(buf))
    cls = json.loads(buf.value[:n].decode())
    lib.mrisr_prof_reset()
    return out, cls


@pytest.mark.parametrize("lora,nk,B", [(True, 77, 2), (False, 77, 2), (True, 16, 1), (True, 80, 3)])
def test_fused_middle_matches_the_separate_launches_and_the_oracle(lora, nk, B):
    import mrisr
    from mrisr import _lib as L
    from oracle import unet as ou
    cfg = _cfg()
    up = ou.init_unet_params(cfg, seed=811, perturb_norm=True)
    p = dict(up)
    if lora:
        p.update(ou.init_lora_params(up, rank=4, seed=812))
    g = torch.Generator().manual_seed(813 + nk)
    x, ctx = torch.randn((B, 4, 16, 16), generator=g), torch.randn((B, nk, 64), generator=g)
    t = torch.tensor([10, 900, 500][:B])
    with torch.no_grad():
        ref = ou.unet_forward(p, cfg, x, t, ctx)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4 if lora else 0, lora_alpha=4)
    net.load_state_dict(p)
    lib = L.lib()
    run = lambda: net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
    try:
        lib.mrisr_debug_xattn_tail(C.c_int(0))
        sep, cls0 = _classes(lib, run)
        lib.mrisr_debug_xattn_tail(C.c_int(1))
        fus, cls1 = _classes(lib, run)
        assert "xattn_tail_c320" not in cls0
        assert cls1.get("xattn_tail_c320", {}).get("launches") == 3, sorted(cls1)   # one block in the encoder, two in the decoder
        assert torch.equal(fus, run()), "not repeatable"
        # every workgroup's LDS pre-filled with NaN bytes: a read that runs ahead of its DMA would show
        lib.mrisr_debug_gemm_flags(C.c_int(2048))
        assert torch.equal(fus, run()), "depends on stale LDS"
    finally:
        lib.mrisr_debug_gemm_flags(C.c_int(0))
        lib.mrisr_debug_xattn_tail(C.c_int(-1))
    assert bool(torch.isfinite(fus.float()).all())
    e_sep, e_fus, d = rel(sep, ref), rel(fus, ref), rel(fus, sep)
    print(f"xtail lora={lora} nk={nk} B={B}: separate vs oracle {e_sep:.3e}, fused vs oracle {e_fus:.3e}, fused vs separate {d:.3e}")
    assert e_sep < 3e-2 and e_fus < 3e-2 and e_fus < 1.3 * e_sep + 2e-3 and d < 2e-2, (e_sep, e_fus, d)


def test_fused_middle_follows_a_new_prompt():
    """The K / V images are re-packed whenever the prompt embedding changes (set_context): two prompts, interleaved."""
    import mrisr
    from oracle import unet as ou
    cfg = _cfg()
    up = ou.init_unet_params(cfg, seed=821, perturb_norm=True)
    g = torch.Generator().manual_seed(822)
    x = torch.randn((2, 4, 16, 16), generator=g)
    c1, c2 = torch.randn((2, 77, 64), generator=g), torch.randn((2, 77, 64), generator=g)
    t = torch.tensor([300, 301])
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16")
    net.load_state_dict(up)
    with torch.no_grad():
        r1, r2 = ou.unet_forward(up, cfg, x, t, c1), ou.unet_forward(up, cfg, x, t, c2)
    o1 = net(x.cuda(), t.cuda(), encoder_hidden_states=c1.cuda()).sample.clone()
    o2 = net(x.cuda(), t.cuda(), encoder_hidden_states=c2.cuda()).sample.clone()
    o1b = net(x.cuda(), t.cuda(), encoder_hidden_states=c1.cuda()).sample.clone()
    assert rel(o1, r1) < 3e-2 and rel(o2, r2) < 3e-2 and torch.equal(o1, o1b)
    assert rel(o1, r2) > 5 * rel(o1, r1)


def test_proj_out_inside_the_fused_feed_forward_matches_its_own_launch():
    """The transformer's proj_out + outer residual as a continuation of the fused feed-forward kernel (its row operand = the feed-forward's
    accumulators, the feed-forward output never stored) against proj_out as its own launch: same rounding points (bf16 after the feed-forward
    + residual, bf16 after the projection, bf16 after the outer residual), so the two agree to accumulation-order noise; both against the oracle."""
    import mrisr
    from mrisr import _lib as L
    from oracle import unet as ou
    cfg = _cfg()
    up = ou.init_unet_params(cfg, seed=831, perturb_norm=True)
    p = dict(up)
    p.update(ou.init_lora_params(up, rank=4, seed=832))
    g = torch.Generator().manual_seed(833)
    x, ctx = torch.randn((2, 4, 16, 16), generator=g), torch.randn((2, 77, 64), generator=g)
    t = torch.tensor([10, 900])
    with torch.no_grad():
        ref = ou.unet_forward(p, cfg, x, t, ctx)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4)
    net.load_state_dict(p)
    lib = L.lib()
    run = lambda: net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
    try:
        lib.mrisr_debug_mlp_proj(C.c_int(0))
        sep, cls0 = _classes(lib, run)
        lib.mrisr_debug_mlp_proj(C.c_int(1))
        fus, cls1 = _classes(lib, run)
        n0 = sum(v["launches"] for k, v in cls0.items() if k.startswith("gemm_bf16_rp320"))
        n1 = sum(v["launches"] for k, v in cls1.items() if k.startswith("gemm_bf16_rp320"))
        assert n0 - n1 == 3, (n0, n1)   # three C = 320 blocks: their proj_out launches are gone
        assert torch.equal(fus, run())
        lib.mrisr_debug_gemm_flags(C.c_int(2048))
        assert torch.equal(fus, run()), "depends on stale LDS"
    finally:
        lib.mrisr_debug_gemm_flags(C.c_int(0))
        lib.mrisr_debug_mlp_proj(C.c_int(-1))
    e_sep, e_fus, d = rel(sep, ref), rel(fus, ref), rel(fus, sep)
    print(f"proj_out in the feed-forward kernel: separate vs oracle {e_sep:.3e}, fused vs oracle {e_fus:.3e}, fused vs separate {d:.3e}")
    assert e_sep < 3e-2 and e_fus < 1.3 * e_sep + 2e-3 and d < 2e-2, (e_sep, e_fus, d)
