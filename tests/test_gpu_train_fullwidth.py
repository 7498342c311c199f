"""GPU: TRAINING parity at FULL SD-1.5 width (VERDICT r2 "what's weak" 3).  The reduced-width tests in test_gpu_train.py check the
backward op by op; here the pieces the 952 samples/s training bench actually runs are composed at their real sizes: flash backward
at head dim 160 inside the model, split-K dgrad convs at 1280 channels, `lora_wgrad` over 320 / 640 / 1280-wide projections, the
zero-stuffed stride-2 dgrad, the one-pass GroupNorm backward on the 8-byte slabs, the 22-projection time-embedding stack with
per-sample timesteps - `mrisr.UNetConfig()` + rank-4 LoRA, B = 2, 4x32x32 latents (256^2 px), the shipped tile table.

Reference being checked against: torch autograd on `oracle.unet.unet_forward` (the notebook's training cell, nb ResDif c11:14-41:
`loss = mse(unet(noisy, t, ehs).sample, noise); backward; clip_grad_norm_(1.0); AdamW.step()`).
Tolerances (north star, written here): f32 engine 1e-3 relative L2 on every adapter gradient tensor and on the flat bucket;
bf16 engine 6e-2 on the flat bucket (bf16 storage of every activation and activation gradient, f32 accumulate)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


@pytest.fixture(scope="module")
def sd15():
    from oracle import unet as ou
    cfg = ou.SD15
    up = ou.init_unet_params(cfg, seed=2101, perturb_norm=True)
    lora = ou.init_lora_params(up, rank=4, seed=2103)
    assert ou.count_params(up) == 859_520_964 and ou.count_params(lora) == 797_184
    return cfg, up, lora


def _batch(seed, B=2, h=32):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, 4, h, h), generator=g)
    ctx = torch.randn((B, 77, 768), generator=g)
    tgt = torch.randn((B, 4, h, h), generator=g)
    t = torch.tensor([37, 912])[:B]
    return x, t, ctx, tgt


@pytest.fixture(scope="module")
def autograd_ref(sd15):
    """One forward + backward of the oracle at full width (a few seconds of host time), shared by the f32 and bf16 checks."""
    from oracle import unet as ou
    cfg, up, lora = sd15
    x, t, ctx, tgt = _batch(2105)
    lp = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
    with torch.enable_grad():
        pred = ou.unet_forward({**up, **lp}, cfg, x, t, ctx, lora_scale=1.0)
        loss = torch.nn.functional.mse_loss(pred, tgt)
        loss.backward()
    return (x, t, ctx, tgt), pred.detach(), float(loss.detach()), {k: v.grad for k, v in lp.items()}


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("bf16", 6e-2)])
def test_sd15_lora_gradients_match_autograd(sd15, autograd_ref, dt, tol):
    import mrisr
    cfg, up, lora = sd15
    (x, t, ctx, tgt), pred_ref, loss_ref, gref = autograd_ref
    net = mrisr.UNet2DConditionModel(mrisr.UNetConfig(), compute_dtype=dt, lora_rank=4, lora_alpha=4, lora_fused=True)
    net.load_state_dict({**up, **lora})
    tr = mrisr.LoRATrainer(net)
    assert tr.num_trainable == 797_184
    tr.zero_grad()
    loss, pred = tr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda(), return_pred=True)
    e_pred = rel(pred, pred_ref)
    flat_ref = torch.cat([gref[k].reshape(-1) for k, _, _ in tr.layout])
    e_flat = rel(tr.grad, flat_ref)
    grads = tr.gradients()
    per = sorted(((rel(grads[k], gref[k]), k) for k in gref), reverse=True)
    print(f"SD-1.5 training [{dt}] B=2 32^2: pred {e_pred:.3e} loss {float(loss):.6f} vs {loss_ref:.6f} bucket {e_flat:.3e} "
          f"worst tensor {per[0][0]:.3e} ({per[0][1]})")
    assert e_pred < (1e-3 if dt == "f32" else 5e-2)
    assert abs(float(loss) - loss_ref) / loss_ref < tol
    assert e_flat < tol, e_flat
    if dt == "f32":
        assert per[0][0] < 1e-3, per[:4]
        # every level's attention kind is in the bucket with a non-trivial gradient (nothing silently skipped)
        for frag in ("down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q", "down_blocks.1.attentions.1.transformer_blocks.0.attn2.to_k",
                     "down_blocks.2.attentions.0.transformer_blocks.0.attn1.to_v", "mid_block.attentions.0.transformer_blocks.0.attn2.to_out.0",
                     "up_blocks.1.attentions.2.transformer_blocks.0.attn1.to_out.0", "up_blocks.3.attentions.0.transformer_blocks.0.attn2.to_q"):
            k = frag + ".lora_B.default.weight"
            assert float(gref[k].norm()) > 0 and rel(grads[k], gref[k]) < 1e-3, k
    else:
        # bf16: no tensor of the bucket is garbage (a wrong kernel variant shows up as O(1))
        assert per[0][0] < 0.35, per[:4]


def test_sd15_clip_adamw_step_matches_torch(sd15):
    """One full optimiser step at full width with the clip active (nb ResDif c11:29-34)."""
    import mrisr
    from oracle import unet as ou
    cfg, up, lora = sd15
    x, t, ctx, tgt = _batch(2107)
    tgt = 500.0 * tgt  # large loss -> the clip is active (gradient norm ~4)
    lp = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
    opt = torch.optim.AdamW(list(lp.values()), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    with torch.enable_grad():
        loss_ref = torch.nn.functional.mse_loss(ou.unet_forward({**up, **lp}, cfg, x, t, ctx, lora_scale=1.0), tgt)
        loss_ref.backward()
    norm_ref = float(torch.nn.utils.clip_grad_norm_(list(lp.values()), 1.0))
    opt.step()
    net = mrisr.UNet2DConditionModel(mrisr.UNetConfig(), compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
    net.load_state_dict({**up, **lora})
    tr = mrisr.LoRATrainer(net, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0)
    loss = tr.step(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda())
    assert abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-3
    assert norm_ref > 1.0 and abs(tr.grad_norm() - norm_ref) / norm_ref < 1e-3, (tr.grad_norm(), norm_ref)
    sd = tr.state_dict()
    worst = max((rel(sd[k], lp[k]), k) for k in lp)
    assert worst[0] < 1e-3, worst
    # and the moved adapters are what the next forward on the handle uses
    x2, t2, ctx2, _ = _batch(2109)
    with torch.no_grad():
        ref = ou.unet_forward({**up, **{k: v.detach() for k, v in lp.items()}}, cfg, x2, t2, ctx2, lora_scale=1.0)
    assert rel(net(x2.cuda(), t2.cuda(), encoder_hidden_states=ctx2.cuda()).sample, ref) < 1e-3


def test_sd15_adapter_xl_gradients_match_autograd(sd15):
    """BASELINE config 3 at its real size: the full `Adapter_XL(sk=True)` (233,743,360 parameters, reference
    src/adapters/modules.py:114-157) on 256^2-px conditions feeding the full-width UNet; d(loss)/d(feature i) - feature 3 through
    the mid block AND skip 11 - and a handful of adapter tensors from every level against autograd."""
    import mrisr
    from oracle import adapter as oa
    from oracle import unet as ou
    cfg, up, lora = sd15
    acfg = oa.ADAPTER_SD15
    ap = oa.init_adapter_params(acfg, seed=2111)
    assert sum(v.numel() for v in ap.values()) == 233_743_360
    B = 2
    x, t, ctx, tgt = _batch(2113)
    cond = torch.randn((B, 3, 256, 256), generator=torch.Generator().manual_seed(2115))
    check = ["conv_in.weight", "body.0.block1.weight", "body.2.block2.bias", "body.3.in_conv.weight", "body.5.block2.weight",
             "body.6.in_conv.weight", "body.8.block1.bias", "body.9.down_opt.op.weight", "body.9.block1.weight", "body.11.block2.weight", "body.11.block2.bias"]
    check = [k for k in check if k in ap]
    assert len(check) >= 8, check
    app = {k: (v.clone().requires_grad_(True) if k in check else v) for k, v in ap.items()}
    with torch.enable_grad():
        feats = oa.adapter_forward(app, acfg, cond)
        # cut the graph at the features: fd[i].grad = d(loss)/d(feature i) through the UNet alone (what the UNet step exports);
        # the adapter's backward continues from there (feature i also feeds the adapter's next stage)
        fd = [f.detach().requires_grad_(True) for f in feats]
        pred = ou.unet_forward({**up, **lora}, cfg, x, t, ctx, down_intrablock_additional_residuals=fd, lora_scale=1.0)
        loss_ref = torch.nn.functional.mse_loss(pred, tgt)
        loss_ref.backward()
        torch.autograd.backward(feats, [f.grad for f in fd])
    assert [tuple(f.shape[1:]) for f in feats] == [(320, 32, 32), (640, 16, 16), (1280, 8, 8), (1280, 4, 4)]
    net = mrisr.UNet2DConditionModel(mrisr.UNetConfig(), compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
    net.load_state_dict({**up, **lora})
    ad = mrisr.Adapter_XL(channels=acfg.channels, nums_rb=acfg.nums_rb, cin=acfg.cin, ksize=acfg.ksize, sk=True, use_conv=True,
                          compute_dtype="f32")
    ad.load_state_dict(ap)
    ltr, atr = mrisr.LoRATrainer(net), mrisr.AdapterTrainer(ad)
    assert atr.num_trainable == 233_743_360
    dfeats = atr.forward(cond.cuda())
    for i, (f, r) in enumerate(zip(dfeats, feats)):
        assert rel(f, r) < 1e-3, (i, rel(f, r))
    fg = atr.new_feature_grads()
    loss = ltr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda(), down_intrablock_additional_residuals=dfeats, feature_grads=fg)
    assert abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-3
    for i, (gf, f) in enumerate(zip(fg, fd)):
        e = rel(gf, f.grad)
        print(f"d(loss)/d(feature {i}): rel {e:.3e}")
        assert e < 1e-3, (i, e)
    atr.backward(fg)
    g = atr.gradients()
    # The adapter is a ReLU network (modules.py:52-110): its backward multiplies by the masks (x > 0) of the forward activations, and
    # two f32 forwards that agree to ~1e-6 disagree on the mask of the few activations that close to zero.  A fraction p of flipped
    # mask bits moves a gradient tensor by ~sqrt(p) in relative L2 - 1e-7...1e-6 of 5.2 M activations per sample is 3e-4...1e-3,
    # growing towards the input as the masks of more layers are crossed.  So: the smooth part of the chain (the UNet: SiLU / GELU,
    # the feature gradients above) is held to 1e-3; the adapter's own tensors to 4e-3 each and 1e-3 in the median.
    errs = {}
    for k in check:
        errs[k] = rel(g[k], app[k].grad)
        print(f"adapter grad {k}: rel {errs[k]:.3e}")
    assert max(errs.values()) < 4e-3, errs
    assert sorted(errs.values())[len(errs) // 2] < 1e-3, errs
