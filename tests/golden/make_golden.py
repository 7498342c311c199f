#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE'S OWN MODULES in the build container.

Run (build container only; /root/reference does not exist on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What comes from where
---------------------
* the arithmetic under test is the reference's: ``src.adapters.res_srdiff`` (forward shift,
  condition image, the whole ``log_validation`` sampler, ``decode_to_vis``) and
  ``src.adapters.modules.Adapter_XL`` - imported from /root/reference, never copied;
* the UNet / ControlNet objects that ``log_validation`` drives are the build's oracle restatement at
  reduced width (oracle.unet.TINY), because diffusers is not installable offline;
* the VAE is a deterministic stub (avg-pool-8 encode, nearest-x8 decode), as in SURVEY.md App. D.3.

Only data (inputs / outputs / seeds / checksums) is written - no reference source.
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from src.adapters import modules as ref_modules  # noqa: E402  (reference)
from src.adapters import res_srdiff as ref  # noqa: E402  (reference)

from oracle import adapter as oad  # noqa: E402
from oracle import schedulers as osch  # noqa: E402
from oracle import unet as ou  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def checksum(params) -> float:
    return float(sum(v.double().abs().sum() for v in params.values()))


class StubVAE:
    """Deterministic VAE stand-in: encode = avgpool8 of channel 0 tiled to 4 ch; decode = nearest x8 of the
    channel mean, 1 channel.  scaling_factor as SD-1.5."""

    class config:
        scaling_factor = 0.18215

    class _Dist:
        def __init__(self, z):
            self.z = z

        def sample(self):
            return self.z

    class _Enc:
        def __init__(self, z):
            self.latent_dist = StubVAE._Dist(z)

    class _Dec:
        def __init__(self, x):
            self.sample = x

    def encode(self, x):
        z = torch.nn.functional.avg_pool2d(x[:, :1], 8).repeat(1, 4, 1, 1)
        return StubVAE._Enc(z)

    def decode(self, z):
        return StubVAE._Dec(torch.nn.functional.interpolate(z.mean(1, keepdim=True), scale_factor=8.0, mode="nearest"))


class Accel:
    device = torch.device("cpu")


def phantom(seed: int, size: int):
    """Smooth seeded test slice in [-1,1] (not the bench phantom; just deterministic content)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn((1, 1, size // 16, size // 16), generator=g)
    img = torch.nn.functional.interpolate(base, size=(size, size), mode="bicubic", align_corners=False)
    return img.clamp(-1, 1)


def gen_forward_shift():
    sched = osch.OracleScheduler()
    g = torch.Generator().manual_seed(11)
    hr = torch.randn((3, 4, 8, 8), generator=g)
    lr = torch.randn((3, 4, 8, 8), generator=g)
    noise = torch.randn((3, 4, 8, 8), generator=g)
    t_scalar = torch.tensor(801)
    t_batch = torch.tensor([0, 500, 999])
    out_s = ref.get_res_shifting_latents(hr, lr, t_scalar, sched, noise)
    out_b = ref.get_res_shifting_latents(hr, lr, t_batch, sched, noise)
    np.savez_compressed(os.path.join(HERE, "res_shift_forward.npz"), hr=hr.numpy(), lr=lr.numpy(),
                        noise=noise.numpy(), t_scalar=t_scalar.numpy(), t_batch=t_batch.numpy(),
                        alphas_cumprod=sched.alphas_cumprod.numpy(), out_scalar=out_s.numpy(),
                        out_batch=out_b.numpy())
    print("res_shift_forward ok")


def gen_condition_and_vis():
    g = torch.Generator().manual_seed(12)
    img = torch.randn((2, 1, 24, 24), generator=g)
    cond = ref.prepare_condition_image(img, target_size=(64, 64))
    cond3 = ref.prepare_condition_image(torch.randn((1, 3, 16, 16), generator=g), target_size=(16, 16))
    dec = torch.randn((2, 1, 16, 16), generator=g) * 1.5
    vis = ref.decode_to_vis(dec, None, is_latent=False)
    np.savez_compressed(os.path.join(HERE, "condition_and_vis.npz"), img=img.numpy(), cond=cond.numpy(),
                        cond3_shape=np.array(cond3.shape), dec=dec.numpy(), vis=vis)
    print("condition_and_vis ok")


def gen_log_validation(n_steps: int, seed: int, tag: str, keep_traj: bool):
    cfg = ou.TINY
    up = ou.init_unet_params(cfg, seed=101, perturb_norm=True)
    up.update(ou.init_lora_params(up, rank=4, seed=103))
    cp = ou.init_controlnet_params(cfg, seed=102, perturb_norm=True)
    unet = ou.OracleUNet(up, cfg)
    unet.record = True
    cnet = ou.OracleControlNet(cp, cfg)
    sched = osch.OracleScheduler(timestep_spacing="leading", steps_offset=1)
    hr = phantom(201, 512)
    lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 4), scale_factor=4.0, mode="bilinear")
    loader = [{"hr": hr, "lr": lr}]
    gctx = torch.Generator().manual_seed(301)
    ctx = torch.randn((1, 77, cfg.cross_attention_dim), generator=gctx)
    torch.manual_seed(seed)
    panel = ref.log_validation(unet, cnet, StubVAE(), loader, sched, torch.float32, Accel(), ctx,
                               num_inference_steps=n_steps)
    panel = np.asarray(panel)
    states = torch.stack(unet.calls)  # state before every step
    W = panel.shape[1] // 3
    gen = panel[:, W:2 * W]
    out = dict(n_steps=np.array(n_steps), seed=np.array(seed), timesteps=sched.timesteps.numpy(),
               unet_checksum=np.array(checksum(up)), controlnet_checksum=np.array(checksum(cp)),
               first_state=states[0].numpy(), last_state_before_final_step=states[-1].numpy(),
               panel_shape=np.array(panel.shape), gen_panel_sha256=np.array(hashlib.sha256(gen.tobytes()).hexdigest()),
               gen_panel_small=gen[::8, ::8, 0].copy())
    if keep_traj:
        out["states"] = states.numpy()
    np.savez_compressed(os.path.join(HERE, f"log_validation_{tag}.npz"), **out)
    print(f"log_validation_{tag} ok: steps={n_steps} states={tuple(states.shape)} panel={panel.shape}")


def gen_adapter():
    cfg = oad.ADAPTER_TINY
    p = oad.init_adapter_params(cfg, seed=401)
    m = ref_modules.Adapter_XL(channels=list(cfg.channels), nums_rb=cfg.nums_rb, cin=cfg.cin, ksize=cfg.ksize,
                               sk=True, use_conv=cfg.use_conv)
    missing = m.load_state_dict(p, strict=True)
    g = torch.Generator().manual_seed(402)
    x = torch.randn((2, 3, 64, 64), generator=g)
    feats = m(x)
    out = {f"feat{i}": f.numpy() for i, f in enumerate(feats)}
    # full-size SD-1.5 adapter: parameter count + output norms at 256^2 px (SURVEY.md 8c iii)
    cfg_full = oad.ADAPTER_SD15
    pf = oad.init_adapter_params(cfg_full, seed=403)
    mf = ref_modules.Adapter_XL(sk=True)
    mf.load_state_dict(pf, strict=True)
    xf = torch.randn((1, 3, 256, 256), generator=torch.Generator().manual_seed(404))  # regenerated in the test
    ff = mf(xf)
    np.savez_compressed(os.path.join(HERE, "adapter_xl.npz"), x=x.numpy(), checksum=np.array(checksum(p)),
                        full_param_count=np.array(sum(v.numel() for v in mf.state_dict().values())),
                        full_checksum=np.array(checksum(pf)),
                        full_norms=np.array([float(f.double().norm()) for f in ff]),
                        full_shapes=np.array([list(f.shape) for f in ff]), **out)
    print("adapter ok", missing, [tuple(f.shape) for f in feats])


def gen_prompts():
    """the reference's prompt producers (utils.py:117-160, res_srdiff.py:125-130) driven with the stub tokenizer /
    text encoder of oracle/prompt_stubs.py, and the reference's log_configs (utils.py:37-71) on the notebook's own config cell."""
    import json
    import random

    import yaml
    from src.adapters import utils as ref_utils  # (reference)

    from oracle.prompt_stubs import StubTextEncoder, StubTokenizer

    sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
    from mrisr.config import TrainConfig

    tok, enc = StubTokenizer(), StubTextEncoder(dim=768, seed=501)
    fixed = ref.get_fixed_prompt_embeds(tok, enc, Accel())
    batch = {"txt": ["high quality MRI scan, T2w brain slice, 3T", ["axial T1w", "sagittal T1w", "coronal T1w"],
                     np.array(["low field 64mT", "high field 3T"]), "high quality mri scan", ["a", "b"], "medical mri scan"]}
    outs = {}
    for tag, p_empty, is_train, seed in (("train", 0.5, True, 7), ("eval", 0.0, False, 8), ("dropall", 1.0, True, 9)):
        tok.seen.clear()
        random.seed(seed)
        e = ref_utils.compute_embeddings_sd1x5(batch, p_empty, [enc], [tok], torch.device("cpu"), is_train=is_train)
        assert set(e) == {"prompt_embeds"}
        pe = e["prompt_embeds"]
        assert tuple(pe.shape) == (6, 77, 768)
        outs[f"{tag}_embeds_small"] = pe[:, ::4, ::24].numpy().copy()  # the stubs are deterministic: a sample + checksums pin it
        outs[f"{tag}_embeds_abs_sum"] = np.array(float(pe.double().abs().sum()))
        outs[f"{tag}_embeds_sha256"] = np.array(hashlib.sha256(pe.numpy().tobytes()).hexdigest())
        outs[f"{tag}_captions"] = np.array(json.dumps(tok.seen[-1]))
        outs[f"{tag}_args"] = np.array(json.dumps({"proportion_empty_prompts": p_empty, "is_train": is_train, "seed": seed}))
        outs[f"{tag}_next_random"] = np.array(random.random())  # the global stream position after the call
    nb = json.load(open("/root/reference/notebooks/ResDif_execution.ipynb"))
    c11 = yaml.safe_load("".join(nb["cells"][11]["source"]).split("\n", 1)[1])  # drop the %%writefile line
    logged = ref_utils.log_configs(TrainConfig.from_dict(c11))
    np.savez_compressed(os.path.join(HERE, "prompt_embeds.npz"), fixed=fixed.numpy(), batch_json=np.array(json.dumps(
        {"txt": [c if isinstance(c, str) else list(map(str, c)) for c in batch["txt"]], "ndarray_items": [2]})),
        c11_config_json=np.array(json.dumps(c11)), log_configs_json=np.array(json.dumps(logged)),
        log_configs_keys=np.array(json.dumps(list(logged))), **outs)
    print("prompts ok", fixed.shape, {k: v.shape for k, v in outs.items() if k.endswith("_small")}, len(c11), "config keys")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "prompts":
        gen_prompts()
        sys.exit(0)
    gen_prompts()
    gen_forward_shift()
    gen_condition_and_vis()
    gen_adapter()
    gen_log_validation(5, 1234, "n5", keep_traj=True)
    gen_log_validation(20, 4321, "n20", keep_traj=False)
