"""CPU: pin the oracle restatements against the golden vectors produced by importing the reference
(tests/golden/make_golden.py).  Nothing here touches /root/reference or a GPU."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import adapter as oad
from oracle import sampler as osa
from oracle import schedulers as osch
from oracle import unet as ou

torch.set_grad_enabled(False)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def checksum(params) -> float:
    return float(sum(v.double().abs().sum() for v in params.values()))


def test_forward_shift_matches_reference(golden_dir):
    g = _load(golden_dir, "res_shift_forward.npz")
    ac = torch.from_numpy(g["alphas_cumprod"])
    hr, lr, nz = (torch.from_numpy(g[k]) for k in ("hr", "lr", "noise"))
    out_s = osa.res_shift_forward(hr, lr, torch.from_numpy(g["t_scalar"]), ac, nz)
    out_b = osa.res_shift_forward(hr, lr, torch.from_numpy(g["t_batch"]), ac, nz)
    assert np.array_equal(out_s.numpy(), g["out_scalar"])  # same op order -> bit exact
    assert np.array_equal(out_b.numpy(), g["out_batch"])
    # and the oracle's own table is the one the fixture was made with
    assert np.array_equal(osch.OracleScheduler().alphas_cumprod.numpy(), g["alphas_cumprod"])


def test_condition_image_and_uint8_panel(golden_dir):
    g = _load(golden_dir, "condition_and_vis.npz")
    cond = osa.condition_image(torch.from_numpy(g["img"]), target_size=(64, 64))
    assert np.array_equal(cond.numpy(), g["cond"])
    assert cond.shape == (2, 3, 64, 64)
    vis = osa.to_uint8_panel(torch.from_numpy(g["dec"]))
    assert vis.dtype == np.uint8 and np.array_equal(vis, g["vis"])


def test_adapter_tiny_matches_reference(golden_dir):
    g = _load(golden_dir, "adapter_xl.npz")
    cfg = oad.ADAPTER_TINY
    p = oad.init_adapter_params(cfg, seed=401)
    assert checksum(p) == pytest.approx(float(g["checksum"]), rel=1e-12)
    feats = oad.adapter_forward(p, cfg, torch.from_numpy(g["x"]))
    for i, f in enumerate(feats):
        np.testing.assert_allclose(f.numpy(), g[f"feat{i}"], rtol=1e-5, atol=1e-6)


def test_adapter_full_param_count_and_norms(golden_dir):
    g = _load(golden_dir, "adapter_xl.npz")
    cfg = oad.ADAPTER_SD15
    p = oad.init_adapter_params(cfg, seed=403)
    assert sum(v.numel() for v in p.values()) == int(g["full_param_count"]) == 233_743_360
    x = torch.randn((1, 3, 256, 256), generator=torch.Generator().manual_seed(404))
    feats = oad.adapter_forward(p, cfg, x)
    assert [list(f.shape) for f in feats] == g["full_shapes"].tolist()
    np.testing.assert_allclose([float(f.double().norm()) for f in feats], g["full_norms"], rtol=1e-4)


def _tiny_models():
    cfg = ou.TINY
    up = ou.init_unet_params(cfg, seed=101, perturb_norm=True)
    up.update(ou.init_lora_params(up, rank=4, seed=103))
    cp = ou.init_controlnet_params(cfg, seed=102, perturb_norm=True)
    return cfg, up, cp


@pytest.mark.parametrize("tag", ["n5", "n20"])
def test_sampler_trajectory_matches_reference_log_validation(golden_dir, tag):
    """The oracle's sampler loop reproduces the reference's log_validation (res_srdiff.py:35-105)
    state-for-state when fed the same noise sequence."""
    g = _load(golden_dir, f"log_validation_{tag}.npz")
    cfg, up, cp = _tiny_models()
    assert checksum(up) == pytest.approx(float(g["unet_checksum"]), rel=1e-12)
    assert checksum(cp) == pytest.approx(float(g["controlnet_checksum"]), rel=1e-12)
    n = int(g["n_steps"])
    sched = osch.OracleScheduler(timestep_spacing="leading", steps_offset=1)
    sched.set_timesteps(n)
    assert np.array_equal(sched.timesteps.numpy(), g["timesteps"])
    # rebuild inputs exactly as make_golden.py did
    gen = torch.Generator().manual_seed(201)
    base = torch.randn((1, 1, 32, 32), generator=gen)
    hr = torch.nn.functional.interpolate(base, size=(512, 512), mode="bicubic", align_corners=False).clamp(-1, 1)
    lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 4), scale_factor=4.0, mode="bilinear")
    ctx = torch.randn((1, 77, cfg.cross_attention_dim), generator=torch.Generator().manual_seed(301))
    lr_lat = torch.nn.functional.avg_pool2d(lr.expand(-1, 3, -1, -1)[:, :1], 8).repeat(1, 4, 1, 1) * 0.18215
    cond = osa.condition_image(lr)
    torch.manual_seed(int(g["seed"]))
    init_noise = torch.randn(lr_lat.shape)
    step_noise = [torch.randn(lr_lat.shape) for _ in range(n)]
    traj = osa.res_srdiff_sample(ou.OracleUNet(up, cfg), ou.OracleControlNet(cp, cfg), lr_lat, ctx, cond,
                                 sched.timesteps.tolist(), sched.alphas_cumprod, init_noise, step_noise)
    np.testing.assert_allclose(traj[0].numpy(), g["first_state"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(traj[n - 1].numpy(), g["last_state_before_final_step"], rtol=1e-4, atol=1e-4)
    if "states" in g.files:
        for i in range(n):
            np.testing.assert_allclose(traj[i].numpy(), g["states"][i], rtol=1e-4, atol=1e-4)
    # final uint8 panel through the stub decoder
    dec = torch.nn.functional.interpolate((traj[-1] / 0.18215).mean(1, keepdim=True), scale_factor=8.0, mode="nearest")
    panel = osa.to_uint8_panel(dec)
    small = panel[::8, ::8, 0]
    assert np.abs(small.astype(int) - g["gen_panel_small"].astype(int)).max() <= 1
