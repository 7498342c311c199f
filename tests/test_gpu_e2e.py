"""End-to-end chain of every built row on the device (SURVEY.md 8a + 8f): slice files -> FastMRILazyDataset -> batched
degradation in the collate -> the reference's own ``log_validation`` call shape (VAE encode, Res-SRDiff sampling loop with the
UNet, VAE decode, uint8 panel) -> PNG folders -> MRIEvaluator.  Reduced-width random-weight models: the point is that the pieces
compose through the reference's interfaces (dict batches, duck-typed model objects, PNG folder contract) and stay finite."""
import json
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


class _Accel:
    device = torch.device("cuda")


def test_dataset_to_metrics_chain(tmp_path):
    import mrisr
    from mrisr import params as P
    rng = np.random.default_rng(3)
    entries = []
    for k in range(4):
        y, x = np.mgrid[0:420, 0:420].astype(np.float32)
        img = 2000 * (1 + np.sin(x / (9 + k)) * np.cos(y / 11)) + rng.uniform(0, 80, (420, 420))
        f = tmp_path / f"slice{k}.npy"
        np.save(f, img.astype(np.uint16))
        entries.append({"filename": str(f), "instanceNumber": k + 1})
    idx = tmp_path / "index.json"
    idx.write_text(json.dumps({"P0": {"3.0T": {"T2": entries}}}))
    ds = mrisr.FastMRILazyDataset(str(idx), mode="train", target_size=(128, 128), fractions=(1.0, 0.0, 0.0), slice_reader=np.load)
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, collate_fn=ds.collate)
    batch = next(iter(loader))
    assert batch["hr"].shape == (4, 1, 128, 128) and batch["lr"].is_cuda
    assert float(batch["hr"].min()) >= -1e-3 and float(batch["hr"].max()) <= 1 + 1e-3

    dev = torch.device("cuda")
    cfg = mrisr.UNetConfig(block_out_channels=(64, 128), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), layers_per_block=1,
                           attention_head_dim=4, cross_attention_dim=64)
    unet = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16")
    unet.load_state_dict(P.random_state_dict(P.unet_param_shapes(cfg), 5, dev))
    vcfg = mrisr.VAEConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1)
    vae = mrisr.AutoencoderKL(vcfg, compute_dtype="bf16")
    vae.load_state_dict(P.random_state_dict(mrisr.vae_param_shapes(vcfg), 6, dev))
    sched = mrisr.DDPMScheduler()
    embeds = torch.randn((1, 77, 64), device=dev)
    torch.manual_seed(0)
    val = [{"hr": 2 * batch["hr"] - 1, "lr": 2 * batch["lr"] - 1}]  # the sampler's [-1, 1] convention (mri_datasets.py:285-289)
    panel = mrisr.log_validation(unet, None, vae, val, sched, torch.float32, _Accel(), embeds, num_inference_steps=5)
    arr = np.asarray(panel)
    assert arr.dtype == np.uint8 and arr.shape[0] == 128 and arr.shape[1] == 3 * 128

    # PNG folder contract of src/eval: <folder>/<name>.png predictions vs ground truth with the same names
    pred_dir, gt_dir = tmp_path / "pred", tmp_path / "gt"
    pred_dir.mkdir()
    gt_dir.mkdir()
    w = 128
    gen = arr[:, w:2 * w] if arr.ndim == 2 else arr[:, w:2 * w, 0]
    hr = arr[:, 2 * w:] if arr.ndim == 2 else arr[:, 2 * w:, 0]
    Image.fromarray(gen).save(pred_dir / "s0.png")
    Image.fromarray(hr).save(gt_dir / "s0.png")
    res = mrisr.MRIEvaluator().evaluate_folders(str(pred_dir), str(gt_dir))
    assert set(res) >= {"PSNR", "SSIM", "HFEN", "NMSE"}
    assert all(np.isfinite(float(v)) for v in res.values())
