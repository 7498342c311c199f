"""GPU: what the reference's prompt producers return (`get_fixed_prompt_embeds` res_srdiff.py:125-130,
`compute_embeddings_sd1x5` utils.py:149-160) goes into the product exactly as they return it - fp16 or fp32, on the encoder's
device or on the CPU, sliced `[0:1]` - through the UNet forward, the fused validation sampler and the fine-tuning step."""
import random

import numpy as np
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
    up = ou.init_unet_params(cfg, seed=611, perturb_norm=True)
    lora = ou.init_lora_params(up, rank=4, seed=613)
    return cfg, up, lora


class _Accel:
    device = torch.device("cuda")


@pytest.mark.parametrize("enc_dtype", [torch.float16, torch.float32])
def test_fixed_prompt_embeds_feed_the_unet_and_the_validation_sampler(tiny, enc_dtype):
    import mrisr
    from oracle import sampler as osa
    from oracle import schedulers as osch
    from oracle import unet as ou
    from oracle.prompt_stubs import StubTextEncoder, StubTokenizer
    cfg, up, lora = tiny
    tok = StubTokenizer()
    enc = StubTextEncoder(dim=cfg.cross_attention_dim, seed=601, dtype=enc_dtype, device="cuda")  # fp16 text encoder, as under mixed_precision
    fixed = mrisr.get_fixed_prompt_embeds(tok, enc, _Accel())
    assert fixed.is_cuda and fixed.dtype == enc_dtype and tuple(fixed.shape) == (1, 77, cfg.cross_attention_dim)
    ctx_ref = fixed.float().cpu()
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4)
    net.load_state_dict({**up, **lora})
    g = torch.Generator().manual_seed(621)
    x = torch.randn((1, 4, 16, 16), generator=g)
    with torch.no_grad():
        ref = ou.unet_forward({**up, **lora}, cfg, x, torch.tensor(500), ctx_ref)
    out = net(x.cuda(), torch.tensor(500).cuda(), encoder_hidden_states=fixed[0:1]).sample  # the reference's call shape (:75)
    assert rel(out, ref) < 1e-3
    # a CPU-resident embedding (compute_embeddings_sd1x5(..., device="cpu")) is moved on the way in
    out = net(x.cuda(), 500, encoder_hidden_states=ctx_ref).sample
    assert rel(out, ref) < 1e-3
    # the fused sampler with the same object
    so = osch.OracleScheduler(timestep_spacing="leading", steps_offset=1)
    so.set_timesteps(3)
    with torch.no_grad():
        traj = osa.ddim_sample(ou.OracleUNet({**up, **lora}, cfg), x, ctx_ref, so)
    sched = mrisr.DDIMScheduler(timestep_spacing="leading", steps_offset=1)
    sched.set_timesteps(3)
    lat = x.cuda().clone()
    mrisr.Sampler(net, sched, kind="ddim").run(lat, fixed[0:1])
    torch.cuda.synchronize()
    assert rel(lat, traj[-1]) < 1e-3


def test_compute_embeddings_feed_the_training_step(tiny):
    """One LoRA step on a batch whose context comes from `compute_embeddings_sd1x5` with caption dropout (the training cell's
    `prompt_embeds`, nb ResDif c11 / utils.py:149-160): loss and gradients equal autograd on the oracle given the same tensor."""
    import mrisr
    from oracle import unet as ou
    from oracle.prompt_stubs import StubTextEncoder, StubTokenizer
    cfg, up, lora = tiny
    tok = StubTokenizer()
    enc = StubTextEncoder(dim=cfg.cross_attention_dim, seed=602, dtype=torch.float32, device="cuda")
    batch = {"txt": ["high quality MRI scan, T2w brain slice, 3T", ["axial T1w", "sagittal T1w"], np.array(["64mT", "3T"]), "mri"]}
    random.seed(3)  # -> ['', 'sagittal T1w', '3T', 'mri']: one caption dropped, one list choice, one ndarray choice, one string
    emb = mrisr.compute_embeddings_sd1x5(batch, 0.5, [enc], [tok], torch.device("cuda"), is_train=True)["prompt_embeds"]
    assert tuple(emb.shape) == (4, 77, cfg.cross_attention_dim) and "" in tok.seen[-1]  # at least one caption dropped at this seed
    g = torch.Generator().manual_seed(631)
    x, tgt = torch.randn((4, 4, 8, 8), generator=g), torch.randn((4, 4, 8, 8), generator=g)
    t = torch.randint(0, 1000, (4,), generator=g)
    lp = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
    with torch.enable_grad():
        loss_ref = torch.nn.functional.mse_loss(ou.unet_forward({**up, **lp}, cfg, x, t, emb.float().cpu(), lora_scale=1.0), tgt)
        loss_ref.backward()
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=True)
    net.load_state_dict({**up, **lora})
    tr = mrisr.LoRATrainer(net)
    loss = tr.forward_backward(x.cuda(), t.cuda(), emb, tgt.cuda())
    assert abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-3
    flat_ref = torch.cat([lp[k].grad.reshape(-1) for k, _, _ in tr.layout])
    assert rel(tr.grad, flat_ref) < 1e-3
