"""GPU parity of the device slice degradation (csrc/data.hip, through the C ABI) against the very library calls the reference
makes - scipy.ndimage.gaussian_filter and PIL.Image.resize on mode "F" (nb ResDif c22:102-154) - wrapped in oracle/data.py.
Tolerance: 3e-6 absolute on images in [0, 1] (both sides accumulate in double and round to f32 once per pass; the only
freedom is summation order)."""
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
TOL = 3e-6


def _imgs(B, H, W, seed=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    base = 0.5 + 0.5 * np.sin(x / 7.0 + seed) * np.cos(y / 5.0)
    return np.clip(base[None] * rng.uniform(0.3, 1.0, (B, 1, 1)) + rng.normal(0, 0.05, (B, H, W)), 0, 1).astype(np.float32)


@pytest.mark.parametrize("H,W,OH,OW,filt", [
    (400, 400, 512, 512, "lanczos"), (400, 400, 256, 256, "lanczos"), (512, 512, 128, 128, "bicubic"),
    (128, 128, 512, 512, "bicubic"), (320, 260, 512, 384, "lanczos"), (77, 131, 33, 200, "bicubic"),
    (64, 64, 64, 17, "lanczos"), (64, 48, 9, 48, "bicubic"), (5, 7, 40, 3, "lanczos"), (30, 30, 30, 30, "bicubic"),
    (1, 1, 8, 8, "bicubic")])
def test_resize_matches_pillow(H, W, OH, OW, filt):
    from mrisr import datasets as D
    from oracle import data as od
    x = _imgs(3, H, W, seed=H + OW)
    pil = Image.LANCZOS if filt == "lanczos" else Image.BICUBIC
    want = np.stack([od.pil_resize(s, (OW, OH), pil) for s in x])
    got = D.resize_slices(torch.from_numpy(x).cuda(), (OH, OW), D.LANCZOS if filt == "lanczos" else D.BICUBIC).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < TOL, np.abs(got - want).max()


@pytest.mark.parametrize("H,W,sigma", [(512, 512, 2.0), (96, 130, 2.0), (5, 9, 2.0), (3, 3, 1.0), (40, 40, 0.7), (64, 64, 6.0)])
def test_gaussian_blur_matches_scipy(H, W, sigma):
    from scipy.ndimage import gaussian_filter
    from mrisr import datasets as D
    x = _imgs(2, H, W, seed=int(10 * sigma) + H)
    want = np.stack([gaussian_filter(s, sigma=sigma) for s in x])
    got = D.gaussian_blur(torch.from_numpy(x).cuda(), sigma).cpu().numpy()
    assert np.abs(got - want).max() < TOL, np.abs(got - want).max()


@pytest.mark.parametrize("H,W,scale", [(512, 512, 4.0), (256, 256, 4.0), (128, 200, 2.0), (100, 100, 3.0)])
def test_simulate_low_field_matches_the_reference_chain(H, W, scale):
    from mrisr import datasets as D
    from oracle import data as od
    x = _imgs(4, H, W, seed=7)
    want = np.stack([od.simulate_low_res(s, (W, H), scale) for s in x])  # target_size is (width, height) to PIL
    got = D.simulate_low_field(torch.from_numpy(x).cuda()[:, None], scale)
    assert got.shape == (4, 1, H, W)
    assert np.abs(got[:, 0].cpu().numpy() - want).max() < TOL


def test_batch_rows_are_independent_and_leading_dims_kept():
    from mrisr import datasets as D
    x = torch.from_numpy(_imgs(6, 64, 64, seed=3)).cuda().reshape(2, 3, 64, 64)
    full = D.simulate_low_field(x, 4.0)
    assert full.shape == x.shape
    assert torch.equal(full[1, 2], D.simulate_low_field(x[1, 2], 4.0))
    assert torch.equal(D.resize_slices(x, (80, 48))[0, 1], D.resize_slices(x[0, 1], (80, 48)))


def test_fastmri_collate_matches_the_reference_items(tmp_path):
    """The whole notebook item - min-max normalise, centre crop 400, LANCZOS to target, low-field simulation - for a batch with
    two different source sizes, against the per-item reference chain."""
    from mrisr import datasets as D
    from oracle import data as od
    rng = np.random.default_rng(5)
    entries, raws = [], []
    for k, shape in enumerate([(448, 448), (448, 448), (320, 300), (448, 448)]):
        raw = (4000 * _imgs(1, *shape, seed=k)[0] + rng.uniform(0, 50, shape)).astype(np.uint16)
        f = tmp_path / f"s{k}.npy"
        np.save(f, raw)
        raws.append(raw)
        entries.append({"filename": str(f), "instanceNumber": 10 + k})
    idx = tmp_path / "index.json"
    idx.write_text(json.dumps({"P0": {"3.0T": {"T2": entries}}}))
    ds = D.FastMRILazyDataset(str(idx), mode="train", target_size=(256, 256), fractions=(1.0, 0.0, 0.0), slice_reader=np.load)
    assert len(ds) == 4
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, collate_fn=ds.collate)
    batch = next(iter(loader))
    assert set(batch) == {"hr", "lr", "txt", "subject_id", "instance"}
    assert batch["hr"].shape == (4, 1, 256, 256) and batch["hr"].is_cuda and batch["instance"] == [10, 11, 12, 13]
    for k, raw in enumerate(raws):
        hr, lr = od.reference_item(raw, (256, 256), 4.0)
        assert np.abs(batch["hr"][k, 0].cpu().numpy() - hr).max() < TOL
        assert np.abs(batch["lr"][k, 0].cpu().numpy() - lr).max() < TOL
    one = ds.item(2)
    assert one["hr"].shape == (1, 256, 256) and one["instance"] == 12 and torch.equal(one["lr"], batch["lr"][2])


def test_data_op_errors():
    import ctypes as C
    from mrisr import _lib as L
    from mrisr import datasets as D
    x = torch.zeros((1, 8, 8), device="cuda")
    with pytest.raises(L.MrisrError):
        D.resize_slices(x, (4, 4), filter=7)
    with pytest.raises(L.MrisrError):
        D.gaussian_blur(x, 0.0)
    with pytest.raises(L.MrisrError):
        D.gaussian_blur(x, 40.0)                    # radius 160 > 64 taps
    with pytest.raises(L.MrisrError):
        D.simulate_low_field(x, 16.0)               # 8 // 16 == 0
    with pytest.raises(L.MrisrError):
        D.resize_slices(torch.zeros((1, 8, 8)), (4, 4))
    out = torch.empty((1, 4, 4), device="cuda")
    rc = L.lib().mrisr_resize_slices(C.c_void_p(x.data_ptr()), 1, 8, 8, C.c_void_p(out.data_ptr()), 4, 4, 0, C.c_void_p(x.data_ptr()), 16, None)
    assert rc != 0 and b"scratch" in L.lib().mrisr_last_error()
