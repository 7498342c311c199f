"""GPU: a ControlNet with its own parameters trainable (SURVEY.md 3.2 / 8 a11 "controlnet params"; the reference itself only runs a
ControlNet for inference, src/adapters/res_srdiff.py:65-70).  One step = recorded ControlNet forward -> UNet step with the 12 + 1
residuals (exports their gradients) -> ControlNet backward with every conv / linear weight and bias differentiated.  Against torch
autograd through oracle.unet.controlnet_forward + unet_forward: f32 1e-3 per tensor (north star), bf16 bucket bound."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


@pytest.fixture(scope="module")
def tiny():
    from oracle import unet as ou
    cfg = ou.TINY
    up = ou.init_unet_params(cfg, seed=711, perturb_norm=True)
    cp = ou.init_controlnet_params(cfg, seed=712, perturb_norm=True)  # non-zero zero-convs: every gradient is exercised
    return cfg, up, cp


def _batch(cfg, B, h, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, 4, h, h), generator=g)
    ctx = torch.randn((B, 16, cfg.cross_attention_dim), generator=g)
    tgt = torch.randn((B, 4, h, h), generator=g)
    cond = torch.randn((B, 3, 8 * h, 8 * h), generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    return x, t, ctx, tgt, cond


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("bf16", 8e-2)])
def test_controlnet_weight_gradients_match_autograd(tiny, dt, tol):
    import mrisr
    from oracle import unet as ou
    cfg, up, cp = tiny
    B, h = 2, 16
    x, t, ctx, tgt, cond = _batch(cfg, B, h, 720)
    cpg = {k: v.clone().requires_grad_(True) for k, v in cp.items()}
    with torch.enable_grad():
        down, mid = ou.controlnet_forward(cpg, cfg, x, t, ctx, cond)
        pred = ou.unet_forward(up, cfg, x, t, ctx, down, mid)
        loss_ref = torch.nn.functional.mse_loss(pred, tgt)
        loss_ref.backward()
    unet = mrisr.UNet2DConditionModel(cfg, compute_dtype=dt)          # frozen: only its input gradients are needed
    unet.load_state_dict(up)
    cnet = mrisr.ControlNetModel(cfg, compute_dtype=dt)
    cnet.load_state_dict(cp)
    utr = mrisr.LoRATrainer(unet)
    ctr = mrisr.ControlNetTrainer(cnet)
    assert ctr.num_trainable == sum(v.numel() for v in cp.values()) and {k for k, _, _ in ctr.layout} == set(cp)
    for k, v in ctr.state_dict().items():
        assert torch.equal(v.cpu(), cp[k]), k
    dd, dm = ctr.forward(x.cuda(), t.cuda(), ctx.cuda(), cond.cuda())
    for k, (a, r) in enumerate(zip(dd + [dm], list(down) + [mid])):
        assert rel(a, r) < (1e-3 if dt == "f32" else 5e-2), (k, rel(a, r))
    dg = ([torch.zeros_like(d) for d in dd], torch.zeros_like(dm))
    loss = utr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda(), down_block_additional_residuals=dd,
                                mid_block_additional_residual=dm, residual_grads=dg)
    assert abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < tol
    ctr.backward(*dg)
    g = ctr.gradients()
    frozen = set(ctr.frozen)
    errs = {k: rel(g[k], cpg[k].grad) for k in cp if k not in frozen}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print(f"ControlNet training [{dt}]: {len(errs)} tensors differentiated, {len(frozen)} frozen; worst {worst[:3]}")
    if worst[0][1] > tol:
        for k in sorted(errs):
            print(f"   {k:90s} {errs[k]:.3e}  |g| {float(g[k].norm()):.3e} ref {float(cpg[k].grad.norm()):.3e}")
    for k in frozen:  # documented gaps: left at exactly zero, never garbage
        assert float(g[k].abs().max()) == 0.0, k
    assert not frozen, sorted(frozen)[:5]   # every one of the ControlNet's tensors is differentiated
    if dt == "f32":
        assert worst[0][1] < 1e-3, worst
    else:
        flat = torch.cat([g[k].reshape(-1).cpu() for k in errs]), torch.cat([cpg[k].grad.reshape(-1) for k in errs])
        assert rel(*flat) < tol, rel(*flat)
    # one optimiser step moves the differentiated tensors like torch.optim.AdamW (clip active) and leaves the frozen ones alone;
    # the re-packed handle then computes with the NEW weights
    if dt == "f32":
        params = [cpg[k] for k in errs]
        opt = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        ctr.lr, ctr.max_grad_norm = 1e-3, 1.0
        ctr.optimizer_step()
        sd = ctr.state_dict()
        w2 = max((rel(sd[k], cpg[k]), k) for k in errs)
        assert w2[0] < 1e-3, w2
        for k in frozen:
            assert torch.equal(sd[k].cpu(), cp[k]), k
        new = {k: (cpg[k].detach() if k in errs else cp[k]) for k in cp}
        with torch.no_grad():
            d2, m2 = ou.controlnet_forward(new, cfg, x, t, ctx, cond)
        o2, om2 = cnet(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda(), controlnet_cond=cond.cuda(), return_dict=False)
        assert rel(om2, m2) < 1e-3 and max(rel(a, b) for a, b in zip(o2, d2)) < 1e-3


def test_controlnet_whole_step_and_bucket_ranges(tiny):
    """`ControlNetTrainer.step` (forward, UNet step, backward, exchange, clip + AdamW, re-pack) moves the parameters like the same step
    done with torch autograd + torch.optim.AdamW on the oracle, twice in a row (the second step runs on the RE-PACKED weights); the
    bucket ranges tile the gradient vector; and LoRA on the UNet can be trained alongside."""
    import mrisr
    from oracle import unet as ou
    cfg, up, cp = tiny
    B, h = 2, 8
    unet = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32")
    unet.load_state_dict(up)
    cnet = mrisr.ControlNetModel(cfg, compute_dtype="f32")
    cnet.load_state_dict(cp)
    utr = mrisr.LoRATrainer(unet)
    ctr = mrisr.ControlNetTrainer(cnet, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0)
    rng = ctr.bucket_ranges()
    cov = sorted((lo, hi) for _, lo, hi in rng)
    assert cov[0][0] == 0 and cov[-1][1] == ctr.num_trainable and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))
    cpg = {k: v.clone().requires_grad_(True) for k, v in cp.items()}
    opt = torch.optim.AdamW(list(cpg.values()), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    for step in range(2):
        x, t, ctx, tgt, cond = _batch(cfg, B, h, 730 + step)
        tgt = 30.0 * tgt  # large loss: the clip is active
        with torch.enable_grad():
            down, mid = ou.controlnet_forward(cpg, cfg, x, t, ctx, cond)
            loss_ref = torch.nn.functional.mse_loss(ou.unet_forward(up, cfg, x, t, ctx, down, mid), tgt)
            opt.zero_grad()
            loss_ref.backward()
        norm_ref = float(torch.nn.utils.clip_grad_norm_(list(cpg.values()), 1.0))
        opt.step()
        loss = ctr.step(utr, x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda(), cond.cuda())
        assert abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-3, step
        assert norm_ref > 1.0 and abs(ctr.grad_norm() - norm_ref) / norm_ref < 1e-3, (step, ctr.grad_norm(), norm_ref)
        sd = ctr.state_dict()
        worst = max((rel(sd[k], cpg[k]), k) for k in cp)
        assert worst[0] < 1e-3, (step, worst)


def test_controlnet_gradients_at_sd15_width():
    """One step at FULL SD-1.5 width (361,279,120 ControlNet parameters + the frozen 859.5 M UNet; B = 1, 4 x 32 x 32 latents, 256^2-px
    condition): a tensor of every kind against autograd - condition embedding (first, 3-channel layer and a stride-2 one), conv_in,
    convs at 320 / 1280 channels, a downsampler, attention projections (fused QKV sections, context K), the GEGLU projection, norm
    affines, the time-embedding MLP and a per-block projection, zero convs."""
    import mrisr
    from oracle import unet as ou
    cfg = ou.SD15
    up = ou.init_unet_params(cfg, seed=741, perturb_norm=True)
    cp = ou.init_controlnet_params(cfg, seed=742, perturb_norm=True)
    assert sum(v.numel() for v in cp.values()) == 361_279_120
    g = torch.Generator().manual_seed(743)
    B, h = 1, 32
    x, tgt = torch.randn((B, 4, h, h), generator=g), torch.randn((B, 4, h, h), generator=g)
    ctx = torch.randn((B, 77, 768), generator=g)
    cond = torch.randn((B, 3, 8 * h, 8 * h), generator=g)
    t = torch.tensor([417])
    check = ["controlnet_cond_embedding.conv_in.weight", "controlnet_cond_embedding.blocks.1.weight", "controlnet_cond_embedding.conv_out.bias",
             "conv_in.weight", "down_blocks.0.resnets.0.conv1.weight", "down_blocks.0.resnets.1.norm2.weight", "down_blocks.0.downsamplers.0.conv.weight",
             "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_k.weight", "down_blocks.1.attentions.1.transformer_blocks.0.attn2.to_k.weight",
             "down_blocks.1.attentions.0.transformer_blocks.0.ff.net.0.proj.weight", "down_blocks.2.attentions.0.transformer_blocks.0.norm2.bias",
             "down_blocks.2.resnets.0.conv_shortcut.weight", "down_blocks.3.resnets.1.conv2.weight", "down_blocks.3.resnets.0.time_emb_proj.weight",
             "mid_block.attentions.0.proj_out.weight", "time_embedding.linear_1.weight", "time_embedding.linear_2.bias",
             "controlnet_down_blocks.4.weight", "controlnet_mid_block.bias"]
    cpg = {k: (v.clone().requires_grad_(True) if k in check else v) for k, v in cp.items()}
    with torch.enable_grad():
        down, mid = ou.controlnet_forward(cpg, cfg, x, t, ctx, cond)
        loss_ref = torch.nn.functional.mse_loss(ou.unet_forward(up, cfg, x, t, ctx, down, mid), tgt)
        loss_ref.backward()
    unet = mrisr.UNet2DConditionModel(mrisr.UNetConfig(), compute_dtype="f32")
    unet.load_state_dict(up)
    cnet = mrisr.ControlNetModel(mrisr.UNetConfig(), compute_dtype="f32")
    cnet.load_state_dict(cp)
    utr, ctr = mrisr.LoRATrainer(unet), mrisr.ControlNetTrainer(cnet)
    assert ctr.num_trainable == 361_279_120 and not ctr.frozen
    dd, dm = ctr.forward(x.cuda(), t.cuda(), ctx.cuda(), cond.cuda())
    dg = ([torch.zeros_like(d) for d in dd], torch.zeros_like(dm))
    loss = utr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda(), down_block_additional_residuals=dd, mid_block_additional_residual=dm,
                                residual_grads=dg)
    assert abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-3
    ctr.backward(*dg)
    gr = ctr.gradients()
    errs = {k: rel(gr[k], cpg[k].grad) for k in check}
    for k, e in errs.items():
        print(f"SD-1.5 ControlNet gradient {k}: rel {e:.3e}")
    assert max(errs.values()) < 1e-3, errs
