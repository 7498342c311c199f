"""Host logic of the dataset mirrors (no GPU): BIDS walk, pad / crop, the npz cache contract of SliceDataset
(src/datasets/mri_datasets.py:162-338) and the JSON index / subject split of the notebook dataset (nb ResDif c22:14-85)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))

from mrisr import datasets as D  # noqa: E402
from mrisr._lib import MrisrError  # noqa: E402


def test_pad_or_center_crop_matches_the_reference_rules():
    x = torch.arange(6.0).reshape(2, 3)
    y = D.pad_or_center_crop(x)
    assert y.shape == (512, 512)
    assert torch.equal(y[255:257, 254:257], x)          # pad_top = 510 // 2, pad_left = 509 // 2
    assert float(y[0, 0]) == -1.0 and float(y.sum()) == float(x.sum()) - (512 * 512 - 6)
    big = torch.arange(600.0 * 520).reshape(600, 520)
    c = D.pad_or_center_crop(big)
    assert torch.equal(c, big[44:556, 4:516])
    tall = torch.zeros((700, 100))
    t = D.pad_or_center_crop(tall, pad_value=5.0)
    assert t.shape == (512, 512) and float(t[:, :206].min()) == 5.0 and float(t[:, 206:306].max()) == 0.0


def test_get_data_dicts_artificial_walks_bids(tmp_path):
    for sid, name in (("sub-01", "sub-01_T2w.nii.gz"), ("sub-02", "sub-02_T1w.nii.gz"), ("sub-03", "sub-03_acq_T2w.nii")):
        d = tmp_path / "rawdata_BIDS_3T" / sid / "anat"
        d.mkdir(parents=True)
        (d / name).write_bytes(b"")
    got = sorted(D.get_data_dicts_artificial(tmp_path), key=lambda r: r["subject_id"])
    assert [g["subject_id"] for g in got] == ["sub-01", "sub-03"]
    assert all(g["hr"] == g["lr"] and "T2w brain slice, 3T field strength" in g["txt"] for g in got)


def _pairs(n):
    return [{"hr": f"/x/{i}.nii", "lr": f"/x/{i}.nii", "txt": f"prompt {i}", "subject_id": f"sub-{i:02d}"} for i in range(n)]


def test_slice_dataset_serves_the_npz_caches(tmp_path):
    rng = np.random.default_rng(0)
    vols = {}
    for sid, shape in (("sub-00", (1, 40, 48, 5)), ("sub-15", (1, 8, 8, 2)), ("sub-01", (1, 530, 20, 3))):
        vols[sid] = (rng.uniform(-1, 1, shape).astype(np.float32), rng.uniform(-1, 1, shape).astype(np.float32))
        np.savez_compressed(tmp_path / f"{sid}_resampled.npz", hr=vols[sid][0], lr=vols[sid][1])
    pairs = [p for p in _pairs(16) if p["subject_id"] in vols]
    ds = D.SliceDataset(pairs, cache_dir=tmp_path)
    assert len(ds) == 5 + 3                                        # sub-15 is skipped (mri_datasets.py:222-224)
    it = ds[2]
    assert set(it) == {"hr", "lr", "txt", "subject_id"} and it["hr"].shape == (1, 512, 512) and it["subject_id"] == "sub-00"
    assert torch.equal(it["hr"][0, 236:276, 232:280], torch.from_numpy(vols["sub-00"][0][0, :, :, 2]))
    assert float(it["hr"][0, 0, 0]) == -1.0
    last = ds[7]
    assert last["subject_id"] == "sub-01" and torch.equal(last["lr"][0, :, 246:266], torch.from_numpy(vols["sub-01"][1][0, 9:521, :, 2]))
    cor = D.SliceDataset(pairs[:1], slice_axis=0, cache_dir=tmp_path)
    assert len(cor) == 40 and cor[3]["hr"].shape == (1, 512, 512)


def test_slice_dataset_builds_a_cache_from_a_reader(tmp_path):
    vol = np.linspace(0, 1800, 20 * 24 * 130, dtype=np.float32).reshape(20, 24, 130)
    ds = D.SliceDataset(_pairs(1), cache_dir=tmp_path, do_registration=False, volume_reader=lambda p: vol)
    assert len(ds) == 130 - 80 - 30                                # crop 80 / -30 along the slice axis (mri_datasets.py:262-274)
    with np.load(tmp_path / "sub-00_resampled.npz") as z:
        assert z["hr"].shape == (1, 20, 24, 20)
        assert np.allclose(z["hr"], np.clip(vol[None, :, :, 80:100] / 900.0, 0, 1) * 2 - 1)
        assert np.allclose(z["lr"], np.clip(vol[None, :, :, 80:100] / 2000.0, 0, 1) * 2 - 1)
    again = D.SliceDataset(_pairs(1), cache_dir=tmp_path)          # second time: served from the cache, no reader needed
    assert torch.equal(again[5]["hr"], ds[5]["hr"])
    small = D.SliceDataset(_pairs(2)[1:], cache_dir=tmp_path, do_registration=False, volume_reader=lambda p: vol[:, :, :50])
    assert len(small) == 50                                        # too few slices: not cropped


def test_slice_dataset_errors(tmp_path):
    with pytest.raises(ValueError):
        D.SliceDataset([], cache_dir=tmp_path)
    with pytest.raises(MrisrError):
        D.SliceDataset(_pairs(1), cache_dir=tmp_path)              # no cache, no reader
    with pytest.raises(MrisrError):
        D.SliceDataset(_pairs(1), cache_dir=tmp_path, do_n4=True, volume_reader=lambda p: np.zeros((4, 4, 4)))
    with pytest.raises(ValueError):
        D.SliceDataset(_pairs(1), cache_dir=tmp_path, do_registration=False, volume_reader=lambda p: np.zeros((4, 4)))


def _index(tmp_path, n_subjects=10, slices=3, shape=(40, 36)):
    rng = np.random.default_rng(1)
    rec = {}
    for s in range(n_subjects):
        entries = []
        for k in range(slices):
            f = tmp_path / f"p{s}_{k}.npy"
            np.save(f, (rng.uniform(0, 4000, shape)).astype(np.uint16))
            entries.append({"filename": str(f), "instanceNumber": k + 1})
        rec[f"P{s}"] = {"3.0T": {"T2": entries}} if s != 4 else {"1.5T": {"T2": entries}}
    path = tmp_path / "index.json"
    path.write_text(json.dumps(rec))
    return path


def test_fastmri_index_and_split(tmp_path):
    path = _index(tmp_path)
    parts = {m: D.FastMRILazyDataset(str(path), mode=m, slice_reader=np.load, device="cpu") for m in ("train", "val", "test")}
    ids = {m: {s["subject_id"] for s in d.subjects} for m, d in parts.items()}
    assert sum(len(v) for v in ids.values()) == 9 and not (ids["train"] & ids["val"]) and not (ids["train"] & ids["test"])
    assert "P4" not in set().union(*ids.values())                  # wrong field strength filtered out
    assert len(parts["train"]) == 3 * len(ids["train"])
    again = D.FastMRILazyDataset(str(path), mode="train", slice_reader=np.load, device="cpu")
    assert [s["subject_id"] for s in again.subjects] == [s["subject_id"] for s in parts["train"].subjects]   # seeded split
    it = parts["train"][0]
    assert set(it) == {"hr_crop", "txt", "subject_id", "instance"} and it["hr_crop"].shape == (40, 36)
    assert float(it["hr_crop"].min()) == 0.0 and float(it["hr_crop"].max()) == 1.0
    assert it["txt"] == "high quality T2 brain MRI, 3.0T field strength, medical imaging"
    with pytest.raises(MrisrError):                                # finishing a batch needs the GPU: no CPU fallback
        parts["train"].collate([it])
