"""GPU: the LoRA fine-tuning step (forward + MSE + backward + clip + AdamW) of libmrisr against torch autograd on the CPU
oracle.  f32 path: 1e-3 relative on every adapter gradient; bf16: relative-L2 bound on the bucket."""

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
    up = ou.init_unet_params(cfg, seed=111, perturb_norm=True)
    lora = ou.init_lora_params(up, rank=4, seed=113)
    return cfg, up, lora


def oracle_grads(cfg, up, lora, x, t, ctx, target, lora_scale, intra=None):
    from oracle import unet as ou
    lp = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
    with torch.enable_grad():
        pred = ou.unet_forward({**up, **lp}, cfg, x, t, ctx, down_intrablock_additional_residuals=intra, lora_scale=lora_scale)
        loss = torch.nn.functional.mse_loss(pred, target)
        loss.backward()
    return pred.detach(), float(loss.detach()), {k: v.grad for k, v in lp.items()}


def make_batch(cfg, B, h, seed, L=77):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, 4, h, h), generator=g)
    ctx = torch.randn((B, L, cfg.cross_attention_dim), generator=g)
    tgt = torch.randn((B, 4, h, h), generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    return x, t, ctx, tgt


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("bf16", 6e-2)])
def test_lora_gradients_match_autograd(tiny, dt, tol):
    import mrisr
    from oracle import unet as ou
    cfg, up, lora = tiny
    B, h = 2, 16
    x, t, ctx, tgt = make_batch(cfg, B, h, 21)
    pred_ref, loss_ref, gref = oracle_grads(cfg, up, lora, x, t, ctx, tgt, lora_scale=2.0)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype=dt, lora_rank=4, lora_alpha=8, lora_fused=True)
    net.load_state_dict({**up, **lora})
    tr = mrisr.LoRATrainer(net)
    assert tr.num_trainable == ou.count_params(lora)
    assert {k for k, _, _ in tr.layout} == set(lora)
    # bound theta reproduces the loaded adapters
    for k, v in tr.state_dict().items():
        assert torch.equal(v.cpu(), lora[k])
    tr.zero_grad()
    loss, pred = tr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda(), return_pred=True)
    assert rel(pred, pred_ref) < tol
    assert abs(float(loss) - loss_ref) / loss_ref < tol
    grads = tr.gradients()
    flat_ref = torch.cat([gref[k].reshape(-1) for k, _, _ in tr.layout])
    assert rel(tr.grad, flat_ref) < tol, rel(tr.grad, flat_ref)
    if dt == "f32":
        worst = max((rel(grads[k], gref[k]), k) for k in gref)
        assert worst[0] < 1e-3, worst
    # gradients ACCUMULATE across micro-batches
    tr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda())
    assert rel(tr.grad, 2 * flat_ref) < tol
    # inference on the same handle still works after a training step (workspace re-planned, flash path back on)
    out = net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
    assert rel(out, pred_ref) < (1e-3 if dt == "f32" else 5e-2)


def test_one_pass_groupnorm_forward_and_backward_match_the_two_kernel_path():
    """bf16 fine-tuning gradients with the one-pass GroupNorm kernels (forward and backward; image slab in registers) against
    the two-kernel path on an SD-1.5-shaped two-level UNet (320 / 640 channels: 10 and 20 channels per group, skip-concat
    slabs, SiLU backward inside): same statistics, so loss and the flat LoRA gradient agree to bf16 rounding noise; the
    f32 engine (always two-kernel) anchors both."""
    import ctypes as C
    import mrisr
    from mrisr import _lib as L
    from mrisr import params as P
    cfg = mrisr.UNetConfig(block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                           layers_per_block=1)
    dev = torch.device("cuda")
    sd = P.random_state_dict(P.unet_param_shapes(cfg), 91, dev)
    sd.update(P.random_state_dict(P.lora_param_shapes(cfg, 4), 92, dev))
    g = torch.Generator(device=dev).manual_seed(93)
    B, h = 8, 16
    x = torch.randn((B, 4, h, h), generator=g, device=dev)
    ctx = torch.randn((B, 77, cfg.cross_attention_dim), generator=g, device=dev)
    tgt = torch.randn((B, 4, h, h), generator=g, device=dev)
    t = torch.randint(0, 1000, (B,), generator=g, device=dev)
    lib = L.lib()
    res = {}
    try:
        for name, dt, fused in (("fused", "bf16", 1), ("two", "bf16", 0), ("f32", "f32", 0)):
            lib.mrisr_debug_gn_fused(C.c_int(fused))
            net = mrisr.UNet2DConditionModel(cfg, compute_dtype=dt, lora_rank=4, lora_alpha=4, lora_fused=True)
            net.load_state_dict(sd)
            tr = mrisr.LoRATrainer(net)
            tr.zero_grad()
            loss = tr.forward_backward(x, t, ctx, tgt)
            torch.cuda.synchronize()
            res[name] = (float(loss), tr.grad.detach().float().clone())
    finally:
        lib.mrisr_debug_gn_fused(C.c_int(1))
    e_f, e_t = rel(res["fused"][1], res["f32"][1]), rel(res["two"][1], res["f32"][1])
    print(f"LoRA gradient vs f32 engine: one-pass GroupNorm {e_f:.3e}, two-kernel {e_t:.3e}; between them {rel(res['fused'][1], res['two'][1]):.3e}")
    assert abs(res["fused"][0] - res["two"][0]) / res["two"][0] < 2e-3
    assert e_f < 8e-2 and e_t < 8e-2
    assert e_f < 1.25 * e_t + 2e-3


def test_training_with_adapter_features_and_scalar_timestep(tiny):
    """cfg 3 shape: T2I-Adapter features enter as constants; timestep given as a 0-dim tensor."""
    import mrisr
    cfg, up, lora = tiny
    B, h = 1, 8
    x, _, ctx, tgt = make_batch(cfg, B, h, 22, L=16)
    t = torch.tensor(417)
    g = torch.Generator().manual_seed(5)
    intra = [0.3 * torch.randn((B, c, h >> i, h >> i), generator=g) for i, c in enumerate(cfg.block_out_channels)]
    pred_ref, loss_ref, gref = oracle_grads(cfg, up, lora, x, t, ctx, tgt, 1.0, intra=intra)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
    net.load_state_dict({**up, **lora})
    tr = mrisr.LoRATrainer(net)
    loss, pred = tr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda(),
                                     down_intrablock_additional_residuals=[f.cuda() for f in intra], return_pred=True)
    assert rel(pred, pred_ref) < 1e-3
    grads = tr.gradients()
    worst = max((rel(grads[k], gref[k]), k) for k in gref)
    assert worst[0] < 1e-3, worst


def test_adamw_clip_step_matches_torch(tiny):
    """Three full steps (clip 1.0 + AdamW, reference nb:ResDif c11:29-34) track torch.optim.AdamW on the oracle."""
    import mrisr
    from oracle import unet as ou
    cfg, up, lora = tiny
    B, h = 2, 8
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
    net.load_state_dict({**up, **lora})
    tr = mrisr.LoRATrainer(net, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0)
    lp = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
    opt = torch.optim.AdamW(list(lp.values()), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    for step in range(3):
        x, t, ctx, tgt = make_batch(cfg, B, h, 30 + step, L=16)
        tgt = 200.0 * tgt  # large loss -> the clip is active
        with torch.enable_grad():
            pred = ou.unet_forward({**up, **lp}, cfg, x, t, ctx, lora_scale=1.0)
            loss_ref = torch.nn.functional.mse_loss(pred, tgt)
            opt.zero_grad()
            loss_ref.backward()
        norm_ref = float(torch.nn.utils.clip_grad_norm_(list(lp.values()), 1.0))
        opt.step()
        loss = tr.step(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda())
        assert abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-3
        assert norm_ref > 1.0 and abs(tr.grad_norm() - norm_ref) / norm_ref < 1e-3
        sd = tr.state_dict()
        worst = max((rel(sd[k], lp[k]), k) for k in lp)
        assert worst[0] < 1e-3, (step, worst)
    # the updated adapters are what inference on the handle now uses
    x, t, ctx, _ = make_batch(cfg, B, h, 40, L=16)
    with torch.no_grad():
        ref = ou.unet_forward({**up, **{k: v.detach() for k, v in lp.items()}}, cfg, x, t, ctx, lora_scale=1.0)
    assert rel(net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample, ref) < 1e-3


def test_optimizer_kernels_against_torch():
    from mrisr import _lib as L
    import ctypes as C
    g = torch.Generator().manual_seed(3)
    n = 100003
    p = torch.randn(n, generator=g).cuda()
    gr = (0.01 * torch.randn(n, generator=g)).cuda()
    m, v = torch.zeros(n).cuda(), torch.zeros(n).cuda()
    pr = p.clone().cpu().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=3e-3, weight_decay=0.05)
    ss = torch.zeros(1).cuda()
    lib = L.lib()
    for step in range(1, 4):
        pr.grad = gr.cpu().clone() / 2  # grad_scale = 1/2 (two ranks summed)
        torch.nn.utils.clip_grad_norm_([pr], 0.5)
        opt.step()
        ss.zero_()
        L.check(lib.mrisr_optim_sumsq(C.c_void_p(gr.data_ptr()), C.c_int64(n), C.c_void_p(ss.data_ptr()), L.stream_ptr()))
        assert abs(float(ss) - float((gr.double() ** 2).sum())) / float(ss) < 1e-5
        L.check(lib.mrisr_optim_adamw(C.c_void_p(p.data_ptr()), C.c_void_p(gr.data_ptr()), C.c_void_p(m.data_ptr()),
                                      C.c_void_p(v.data_ptr()), C.c_int64(n), C.c_void_p(ss.data_ptr()), C.c_float(0.5),
                                      C.c_float(0.5), C.c_float(3e-3), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8),
                                      C.c_float(0.05), C.c_int(step), L.stream_ptr()))
        assert rel(p, pr) < 1e-5


def test_train_errors(tiny):
    import mrisr
    cfg, up, lora = tiny
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=False)
    net.load_state_dict({**up, **lora})
    with pytest.raises(mrisr.MrisrError, match="un-merged"):
        mrisr.LoRATrainer(net)
    net2 = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
    net2.load_state_dict({**up, **lora})
    tr = mrisr.LoRATrainer(net2)
    x, t, ctx, tgt = make_batch(cfg, 1, 8, 50, L=16)
    with pytest.raises(mrisr.MrisrError, match="target"):
        tr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt[:, :2].contiguous().cuda())


def test_step_through_rccl_process_group(tiny):
    """The exchange step on the real backend: a (single-rank) RCCL group - the flat gradient bucket is all-reduced on the
    device between backward and the optimiser, and the result equals the group-less step."""
    import socket
    import torch.distributed as dist
    import mrisr
    cfg, up, lora = tiny
    x, t, ctx, tgt = make_batch(cfg, 2, 8, 60, L=16)

    def one_step():
        net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
        net.load_state_dict({**up, **lora})
        tr = mrisr.LoRATrainer(net, lr=1e-3)
        tr.step(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda())
        return tr.theta.clone()

    ref = one_step()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        got = one_step()
    finally:
        dist.destroy_process_group()
    assert rel(got, ref) < 1e-5


# ---- T2I-Adapter training (BASELINE config 3: the adapter is differentiated every step) ----
def _adapter_setup(seed=131):
    from oracle import adapter as oa
    acfg = oa.AdapterConfig(channels=(64, 128, 256, 256), nums_rb=2, cin=192, ksize=3)
    return acfg, oa.init_adapter_params(acfg, seed=seed)


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("bf16", 8e-2)])
def test_adapter_gradients_match_autograd(tiny, dt, tol):
    """d(loss)/d(every adapter conv weight and bias) and d(loss)/d(LoRA) for loss = mse(unet(x, t, ctx, adapter(cond)), eps):
    the UNet step hands the feature gradients to the adapter's backward (dgrad convs + one pixel-contraction GEMM per tap)."""
    import mrisr
    from oracle import adapter as oa
    from oracle import unet as ou
    cfg, up, lora = tiny
    acfg, ap = _adapter_setup()
    B, h = 2, 8
    x, t, ctx, tgt = make_batch(cfg, B, h, 70, L=16)
    cond = torch.randn((B, 3, 8 * h, 8 * h), generator=torch.Generator().manual_seed(71))
    lp = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
    app = {k: v.clone().requires_grad_(True) for k, v in ap.items()}
    with torch.enable_grad():
        feats = oa.adapter_forward(app, acfg, cond)
        # cut the graph at the features: fd[i].grad is then d(loss)/d(feature i) THROUGH THE UNET alone - what the UNet step
        # exports - and the adapter's own backward continues from there (feature i also feeds the adapter's next stage)
        fd = [f.detach().requires_grad_(True) for f in feats]
        pred = ou.unet_forward({**up, **lp}, cfg, x, t, ctx, down_intrablock_additional_residuals=fd, lora_scale=1.0)
        loss_ref = torch.nn.functional.mse_loss(pred, tgt)
        loss_ref.backward()
        torch.autograd.backward(feats, [f.grad for f in fd])
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype=dt, lora_rank=4, lora_alpha=4, lora_fused=True)
    net.load_state_dict({**up, **lora})
    ad = mrisr.Adapter_XL(channels=acfg.channels, nums_rb=acfg.nums_rb, cin=acfg.cin, ksize=acfg.ksize, compute_dtype=dt)
    ad.load_state_dict(ap)
    ltr, atr = mrisr.LoRATrainer(net), mrisr.AdapterTrainer(ad)
    assert atr.num_trainable == sum(v.numel() for v in ap.values()) and {k for k, _, _ in atr.layout} == set(ap)
    for k, v in atr.state_dict().items():
        assert torch.equal(v.cpu(), ap[k])
    dfeats = atr.forward(cond.cuda())
    for f, r in zip(dfeats, feats):
        assert rel(f, r) < (1e-3 if dt == "f32" else 3e-2)
    fg = atr.new_feature_grads()
    loss = ltr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda(), down_intrablock_additional_residuals=dfeats, feature_grads=fg)
    assert abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < tol
    # d(loss)/d(feature i): feature 3's gradient is the mid block's PLUS the decoder's through skip 11 (the in-place add
    # of diffusers' attention-free hand-off lands in res_samples[-1])
    for i, (gf, f) in enumerate(zip(fg, fd)):
        assert rel(gf, f.grad) < tol, (i, rel(gf, f.grad))
    atr.backward(fg)
    flat_ref = torch.cat([app[k].grad.reshape(-1) for k, _, _ in atr.layout])
    assert rel(atr.grad, flat_ref) < tol, rel(atr.grad, flat_ref)
    lflat_ref = torch.cat([lp[k].grad.reshape(-1) for k, _, _ in ltr.layout])
    assert rel(ltr.grad, lflat_ref) < tol
    if dt == "f32":
        g = atr.gradients()
        worst = max((rel(g[k], app[k].grad), k) for k in app)
        assert worst[0] < 1e-3, worst


def test_joint_lora_adapter_step_matches_torch(tiny):
    """Two full config-3 steps (adapter + UNet forward, backward of both, ONE clip over the union, AdamW on both) track
    torch.optim.AdamW + clip_grad_norm_ on the oracle."""
    import mrisr
    from oracle import adapter as oa
    from oracle import unet as ou
    cfg, up, lora = tiny
    acfg, ap = _adapter_setup(seed=141)
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
    net.load_state_dict({**up, **lora})
    ad = mrisr.Adapter_XL(channels=acfg.channels, nums_rb=acfg.nums_rb, cin=acfg.cin, ksize=acfg.ksize, compute_dtype="f32")
    ad.load_state_dict(ap)
    kw = dict(lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0)
    ltr, atr = mrisr.LoRATrainer(net, **kw), mrisr.AdapterTrainer(ad, **kw)
    lp = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
    app = {k: v.clone().requires_grad_(True) for k, v in ap.items()}
    params = list(lp.values()) + list(app.values())
    opt = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    B, h = 2, 8
    for step in range(2):
        x, t, ctx, tgt = make_batch(cfg, B, h, 80 + step, L=16)
        tgt = 100.0 * tgt  # clip active
        cond = torch.randn((B, 3, 8 * h, 8 * h), generator=torch.Generator().manual_seed(90 + step))
        with torch.enable_grad():
            pred = ou.unet_forward({**up, **lp}, cfg, x, t, ctx, down_intrablock_additional_residuals=oa.adapter_forward(app, acfg, cond),
                                   lora_scale=1.0)
            loss_ref = torch.nn.functional.mse_loss(pred, tgt)
            opt.zero_grad()
            loss_ref.backward()
        norm_ref = float(torch.nn.utils.clip_grad_norm_(params, 1.0))
        opt.step()
        loss = mrisr.joint_step(ltr, atr, x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda(), cond.cuda())
        assert abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-3
        assert norm_ref > 1.0 and abs(ltr.grad_norm() - norm_ref) / norm_ref < 1e-3
        for sd, ref in ((ltr.state_dict(), lp), (atr.state_dict(), app)):
            worst = max((rel(sd[k], ref[k]), k) for k in ref)
            assert worst[0] < 1e-3, (step, worst)
    # the updated adapter is what its forward now computes
    cond = torch.randn((1, 3, 64, 64), generator=torch.Generator().manual_seed(99))
    with torch.no_grad():
        ref = oa.adapter_forward({k: v.detach() for k, v in app.items()}, acfg, cond)
    for f, r in zip(ad(cond.cuda()), ref):
        assert rel(f, r) < 1e-3


@pytest.mark.parametrize("mode", ["all_reduce", "reduce_scatter"])
def test_overlapped_joint_step_equals_joint_step(tiny, mode):
    """the config-3 step with the adapter's backward cut at level boundaries and each level's gradient bucket
    handed to the exchange as soon as it is final (mrisr.joint_step_overlapped) gives the same parameters as the round-1 step
    (whole backward, then one all-reduce) - without a process group and through a single-rank RCCL group, both exchange modes;
    level ranges tile the flat vector and the level-wise backward refuses a wrong order."""
    import socket
    import torch.distributed as dist
    import mrisr
    cfg, up, lora = tiny
    acfg, ap = _adapter_setup(seed=151)
    B, h = 2, 8

    def run(step_fn, steps=2):
        net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
        net.load_state_dict({**up, **lora})
        ad = mrisr.Adapter_XL(channels=acfg.channels, nums_rb=acfg.nums_rb, cin=acfg.cin, ksize=acfg.ksize, compute_dtype="f32")
        ad.load_state_dict(ap)
        kw = dict(lr=1e-3, max_grad_norm=1.0)
        ltr, atr = mrisr.LoRATrainer(net, **kw), mrisr.AdapterTrainer(ad, **kw)
        losses = []
        for s in range(steps):
            x, t, ctx, tgt = make_batch(cfg, B, h, 180 + s, L=16)
            cond = torch.randn((B, 3, 8 * h, 8 * h), generator=torch.Generator().manual_seed(190 + s))
            losses.append(float(step_fn(ltr, atr, x.cuda(), t.cuda(), ctx.cuda(), (50.0 * tgt).cuda(), cond.cuda())))
        return ltr, atr, losses

    l0, a0, loss0 = run(mrisr.joint_step)
    # level ranges: contiguous, descending levels tile [0, n)
    rng = [a0.level_range(l) for l in range(a0.num_levels)]
    assert rng[0][0] == 0 and rng[-1][1] == a0.num_trainable and all(rng[i][1] == rng[i + 1][0] for i in range(len(rng) - 1))
    with pytest.raises(mrisr.MrisrError, match="descending"):
        a0.forward(torch.zeros(B, 3, 8 * h, 8 * h).cuda())
        a0.backward_level(a0.new_feature_grads(), 0)
    l1, a1, loss1 = run(lambda *a: mrisr.joint_step_overlapped(*a, mode=mode))
    assert loss1 == loss0 and rel(l1.theta, l0.theta) < 1e-6 and rel(a1.theta, a0.theta) < 1e-6
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        l2, a2, loss2 = run(lambda *a: mrisr.joint_step_overlapped(*a, mode=mode))
    finally:
        dist.destroy_process_group()
    assert rel(l2.theta, l0.theta) < 1e-5 and rel(a2.theta, a0.theta) < 1e-5 and abs(loss2[-1] - loss0[-1]) < 1e-5 * abs(loss0[-1])


def test_ema_and_checkpoint_roundtrip(tiny, tmp_path):
    """EMA of the adapters (diffusers EMAModel.step) and checkpoint interchange: parameters as safetensors under their peft
    keys, optimiser moments alongside; a restored trainer continues bit-identically."""
    import mrisr
    from safetensors.torch import load_file
    cfg, up, lora = tiny
    x, t, ctx, tgt = make_batch(cfg, 2, 8, 120, L=16)

    def fresh():
        net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
        net.load_state_dict({**up, **lora})
        return mrisr.LoRATrainer(net, lr=1e-3)

    tr = fresh()
    theta0 = torch.cat([lora[k].reshape(-1) for k, _, _ in tr.layout]).cuda()
    tr.ema_init()                          # EMAModel(parameters): shadow = the parameters at construction
    tr.step(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda())
    assert tr.ema_step(0.9) == 0.0         # diffusers EMAModel.get_decay: optimisation step 1 -> decay 0 -> shadow = theta
    assert rel(tr.ema, tr.theta) < 1e-7
    theta1 = tr.theta.clone()
    tr.step(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda())
    d = tr.ema_step(0.9)                   # step 2 -> (1 + 1) / (10 + 1)
    assert abs(d - 2.0 / 11.0) < 1e-12
    assert rel(tr.ema, d * theta1 + (1 - d) * tr.theta) < 1e-6
    for _ in range(98):
        tr.ema_steps += 1
    assert tr.ema_decay_at(tr.ema_steps + 1, 0.9) == 0.9   # ... rising to the ceiling
    assert not torch.equal(theta0, tr.theta)
    # unet.state_dict() reports the TRAINED adapters (it used to return the tensors handed to load_state_dict)
    usd = tr.unet.state_dict()
    k0 = tr.layout[0][0]
    assert torch.equal(usd[k0].float().cpu(), tr.state_dict()[k0].cpu()) and not torch.equal(usd[k0].float().cpu(), lora[k0])
    path = str(tmp_path / "lora.safetensors")
    tr.save_checkpoint(path)               # peft's on-disk form: adapter name stripped, "base_model.model." prefix
    sd = load_file(path)
    want = {"base_model.model." + k.replace(".default.", "."): k for k in lora}
    assert set(sd) == set(want) and all(torch.equal(sd[k], tr.state_dict()[want[k]].cpu()) for k in sd)
    tr.save_checkpoint(str(tmp_path / "d.safetensors"), key_format="diffusers")
    assert set(load_file(str(tmp_path / "d.safetensors"))) == {"unet." + k.replace(".default.", ".") for k in lora}
    n_before = tr.step_count
    tr.step(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda())
    tr2 = fresh()
    tr2.load_checkpoint(path)
    assert tr2.step_count == n_before and tr2.ema_steps == tr.ema_steps
    tr2.step(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda())
    assert rel(tr2.theta, tr.theta) < 1e-6
    # a model built straight from the on-disk keys (either form) equals the trained one
    net3 = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
    net3.load_state_dict({**up, **load_file(str(tmp_path / "d.safetensors"))})
    assert set(net3.state_dict()) == set(up) | set(lora)
    with pytest.raises(KeyError):
        bad = dict(sd); bad.pop(next(iter(bad)))
        from safetensors.torch import save_file
        save_file(bad, str(tmp_path / "bad.safetensors"))
        fresh().load_checkpoint(str(tmp_path / "bad.safetensors"))
    tr.save_checkpoint(str(tmp_path / "ema.safetensors"), use_ema=True)
    from mrisr.train import lora_keys_from_disk
    ema_sd = lora_keys_from_disk(load_file(str(tmp_path / "ema.safetensors")))
    assert rel(torch.cat([ema_sd[k].reshape(-1) for k, _, _ in tr.layout]), tr.ema) < 1e-7


@pytest.mark.parametrize("frozen", [True, False])
def test_controlnet_residual_gradients_match_autograd(tiny, frozen):
    """The training graph of the reference's ControlNet configuration (unet(..., down_block_additional_residuals=down_res,
    mid_block_additional_residual=mid_res), res_srdiff.py:73-78): d(loss)/d(each of the 12 + 1 residuals) out of the UNet step -
    the seeds of the ControlNet's own backward - against autograd on the oracle; with a FROZEN UNet (no adapters: inputs only) and
    with LoRA trained alongside (the encoder's saved activations must not see the residuals: diffusers adds them out of place)."""
    import mrisr
    from oracle import unet as ou
    cfg, up, lora = tiny
    B, h = 2, 16
    x, t, ctx, tgt = make_batch(cfg, B, h, 90, L=16)
    g = torch.Generator().manual_seed(91)
    chans, sizes = cfg.skip_channels(), [h, h, h, h // 2, h // 2, h // 2, h // 4, h // 4, h // 4, h // 8, h // 8, h // 8]
    down = [(0.3 * torch.randn((B, c, s, s), generator=g)).requires_grad_(True) for c, s in zip(chans, sizes)]
    mid = (0.3 * torch.randn((B, cfg.block_out_channels[-1], h // 8, h // 8), generator=g)).requires_grad_(True)
    lp = {k: v.clone().requires_grad_(not frozen) for k, v in lora.items()}
    params = dict(up) if frozen else {**up, **lp}
    with torch.enable_grad():
        pred = ou.unet_forward(params, cfg, x, t, ctx, down, mid, lora_scale=1.0)
        loss_ref = torch.nn.functional.mse_loss(pred, tgt)
        loss_ref.backward()
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=0 if frozen else 4, lora_alpha=None if frozen else 4, lora_fused=True)
    net.load_state_dict(params if frozen else {**up, **lora})
    tr = mrisr.LoRATrainer(net)
    assert (tr.num_trainable == 0) == frozen
    dg = ([torch.zeros_like(d).cuda() for d in down], torch.zeros_like(mid).cuda())
    loss, p = tr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda(), return_pred=True,
                                  down_block_additional_residuals=[d.detach().cuda() for d in down],
                                  mid_block_additional_residual=mid.detach().cuda(), residual_grads=dg)
    assert rel(p, pred) < 1e-3 and abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-3
    for k, (a, d) in enumerate(zip(dg[0], down)):
        assert rel(a, d.grad) < 1e-3, (k, rel(a, d.grad))
    assert rel(dg[1], mid.grad) < 1e-3
    if not frozen:
        flat_ref = torch.cat([lp[k].grad.reshape(-1) for k, _, _ in tr.layout])
        assert rel(tr.grad, flat_ref) < 1e-3, rel(tr.grad, flat_ref)
    # and the residual state does not leak into the next plain step
    if frozen:  # nothing is trainable and no input asks for a gradient: a clean error, not a silent no-op
        with pytest.raises(mrisr.MrisrError, match="no trainable"):
            tr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda())
    else:
        l2, p2 = tr.forward_backward(x.cuda(), t.cuda(), ctx.cuda(), tgt.cuda(), return_pred=True)
        with torch.no_grad():
            assert rel(p2, ou.unet_forward({k: v.detach() for k, v in params.items()}, cfg, x, t, ctx)) < 1e-3
