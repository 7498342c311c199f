"""Slice datasets with the reference's batch contract ``{'hr', 'lr', 'txt', 'subject_id'[, 'instance']}``.

* ``SliceDataset`` mirrors ``src/datasets/mri_datasets.py:191-338`` (NIfTI path, values in [-1, 1], 512 x 512 pad / crop): it
  serves the per-subject ``{sid}_resampled.npz`` caches the reference writes.  Volumes that are not cached yet need a
  ``volume_reader(path) -> ndarray [H, W, D]``; SimpleITK's rigid registration / N4 (``mri_datasets.py:45-105``) is outside this
  build, and in the reference's "artificial" pairs HR and LR are the same file anyway (``mri_datasets.py:8-43``).
* ``FastMRILazyDataset`` mirrors the notebook dataset (nb ResDif c22): JSON index, subject-level split, lazy slice reads, values
  in [0, 1].  Its per-slice LANCZOS resize and low-field simulation (scipy gaussian_filter + PIL BICUBIC down / up) run on the
  device for the whole batch (``csrc/data.hip``) inside ``collate``: ``__getitem__`` only reads, normalises and centre-crops.
  Use ``DataLoader(ds, batch_size=B, collate_fn=ds.collate)``; ``ds.item(i)`` gives the reference's single-item dict.

The device functions ``resize_slices`` / ``gaussian_blur`` / ``simulate_low_field`` are exported for other pipelines.  No CPU
fallback: they raise without ``libmrisr.so`` or a GPU."""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset, random_split

from . import _lib as L

BICUBIC, LANCZOS = 0, 1


# ---------------------------------------------------------------------------------------------- device ops
def _slices(x: torch.Tensor) -> Tuple[torch.Tensor, Tuple[int, ...]]:
    if not x.is_cuda:
        raise L.MrisrError("slice degradation runs on the GPU (gfx950); there is no CPU fallback")
    if x.ndim < 2:
        raise ValueError("expected [..., H, W]")
    lead = tuple(x.shape[:-2])
    return x.to(torch.float32).reshape((-1,) + tuple(x.shape[-2:])).contiguous(), lead


def resize_slices(x: torch.Tensor, size_hw: Tuple[int, int], filter: int = LANCZOS) -> torch.Tensor:
    """Pillow ``Image.resize`` on mode "F" for every [H, W] slice of ``x`` [..., H, W]; ``size_hw`` = (out_h, out_w)."""
    x3, lead = _slices(x)
    B, H, W = x3.shape
    OH, OW = int(size_hw[0]), int(size_hw[1])
    if OH < 1 or OW < 1:
        raise ValueError("output size must be positive")
    lib = L.lib()
    nb = lib.mrisr_resize_scratch_bytes(B, H, W, OH, OW, int(filter))
    scratch = torch.empty((nb,), dtype=torch.uint8, device=x3.device)
    out = torch.empty((B, OH, OW), dtype=torch.float32, device=x3.device)
    L.check(lib.mrisr_resize_slices(C.c_void_p(x3.data_ptr()), B, H, W, C.c_void_p(out.data_ptr()), OH, OW, int(filter),
                                    C.c_void_p(scratch.data_ptr()), nb, L.stream_ptr()))
    scratch.record_stream(torch.cuda.current_stream())
    return out.reshape(lead + (OH, OW))


def gaussian_blur(x: torch.Tensor, sigma: float, truncate: float = 4.0) -> torch.Tensor:
    """``scipy.ndimage.gaussian_filter(slice, sigma)`` (mode "reflect") for every slice of ``x`` [..., H, W]."""
    x3, lead = _slices(x)
    B, H, W = x3.shape
    tmp = torch.empty_like(x3)
    out = torch.empty_like(x3)
    L.check(L.lib().mrisr_gaussian_blur_slices(C.c_void_p(x3.data_ptr()), B, H, W, float(sigma), float(truncate),
                                               C.c_void_p(tmp.data_ptr()), C.c_void_p(out.data_ptr()), L.stream_ptr()))
    tmp.record_stream(torch.cuda.current_stream())
    return out.reshape(lead + (H, W))


def simulate_low_field(hr: torch.Tensor, scale_factor: float = 4.0) -> torch.Tensor:
    """nb ResDif c22:140-154 for a batch: blur(sigma = 0.5 * scale) -> BICUBIC down -> BICUBIC back.  The intermediate image
    is (W // s) rows by (H // s) columns, as the reference's tuple order makes it (same thing for square slices)."""
    x3, lead = _slices(hr)
    B, H, W = x3.shape
    lib = L.lib()
    nb = lib.mrisr_low_field_scratch_bytes(B, H, W, float(scale_factor))
    scratch = torch.empty((nb,), dtype=torch.uint8, device=x3.device)
    out = torch.empty_like(x3)
    L.check(lib.mrisr_simulate_low_field(C.c_void_p(x3.data_ptr()), B, H, W, float(scale_factor), C.c_void_p(out.data_ptr()),
                                         C.c_void_p(scratch.data_ptr()), nb, L.stream_ptr()))
    scratch.record_stream(torch.cuda.current_stream())
    return out.reshape(lead + (H, W))


# ---------------------------------------------------------------------------------------------- NIfTI path
def get_data_dicts_artificial(data_dir, modality: str = "T2w") -> List[Dict]:
    """BIDS walk of ``mri_datasets.py:8-43``: one dict per 3T subject, 'hr' and 'lr' both pointing at the 3T volume."""
    base = Path(data_dir) / "rawdata_BIDS_3T"
    out = []
    for subject_dir in base.glob("sub-*"):
        files = list((subject_dir / "anat").glob(f"*{modality}*.nii*"))
        if files:
            path = str(files[0])
            out.append({"lr": path, "hr": path, "subject_id": subject_dir.name,
                        "txt": f"high quality MRI scan, {modality} brain slice, 3T field strength, precise anatomical details, "
                               "sharp focus, medical imaging"})
    return out


def pad_or_center_crop(tensor2d: torch.Tensor, pad_value: float = -1.0, target: Tuple[int, int] = (512, 512)) -> torch.Tensor:
    """``mri_datasets.py:162-188``: centre-crop what exceeds ``target``, pad the rest symmetrically (extra pixel at the end)."""
    th, tw = target
    H, W = tensor2d.shape
    if H > th:
        s = (H - th) // 2
        tensor2d = tensor2d[s:s + th, :]
        H = th
    if W > tw:
        s = (W - tw) // 2
        tensor2d = tensor2d[:, s:s + tw]
        W = tw
    ph, pw = max(0, th - H), max(0, tw - W)
    if ph or pw:
        top, left = ph // 2, pw // 2
        tensor2d = torch.nn.functional.pad(tensor2d.unsqueeze(0), (left, pw - left, top, ph - top), mode="constant",
                                           value=pad_value).squeeze(0)
    return tensor2d


class SliceDataset(Dataset):
    """Slices of cached, intensity-normalised volumes ([1, H, W, D] float32 in [-1, 1]); same constructor arguments and item
    dict as the reference's class.  ``do_registration`` / ``do_n4`` are accepted for signature parity; asking for them on an
    uncached volume raises (SimpleITK is not part of this build)."""

    SKIP_SUBJECTS = ("sub-15",)  # mri_datasets.py:222-224 "wrong layout"

    def __init__(self, pairs: Sequence[Dict], slice_axis: int = 2, cache_dir="./cache", do_registration: bool = True,
                 do_n4: bool = False, lr_clip=(0, 2000), hr_clip=(0, 900),
                 volume_reader: Optional[Callable[[str], np.ndarray]] = None):
        if not pairs:
            raise ValueError("No pairs found. Check paths.")
        if slice_axis not in (0, 1, 2):
            raise ValueError("slice_axis must be 0 (sagittal), 1 (coronal) or 2 (axial)")
        self.pairs, self.slice_axis = list(pairs), slice_axis
        self.cache_dir = Path(cache_dir)
        self.cache_dir.mkdir(parents=True, exist_ok=True)
        self.do_registration, self.do_n4 = do_registration, do_n4
        self.lr_clip, self.hr_clip = lr_clip, hr_clip
        self.volume_reader = volume_reader
        self.slice_metadata: List[Dict] = []
        self._prepare_all_pairs()

    def _load_uncached(self, item):
        if self.volume_reader is None:
            raise L.MrisrError(f"{item['subject_id']}: no cache file and no volume_reader; NIfTI decoding / registration "
                               "(SimpleITK in the reference) is outside this build - pass volume_reader or pre-built caches")
        if self.do_n4 or (self.do_registration and item["hr"] != item["lr"]):
            raise L.MrisrError("rigid registration / N4 bias correction need SimpleITK (mri_datasets.py:45-105); register "
                               "offline and pass do_registration=False")
        hr = np.asarray(self.volume_reader(item["hr"]), dtype=np.float32)
        lr = hr if item["lr"] == item["hr"] else np.asarray(self.volume_reader(item["lr"]), dtype=np.float32)
        if hr.ndim != 3 or lr.shape != hr.shape:
            raise ValueError(f"{item['subject_id']}: volume_reader must return [H, W, D] arrays on one grid")
        hr, lr = hr[None], lr[None]
        dim = self.slice_axis + 1
        n = hr.shape[dim]
        start, end = 80, n - 30  # mri_datasets.py:262-274
        if end > start and n > 60:
            sl = [slice(None)] * 4
            sl[dim] = slice(start, end)
            hr, lr = hr[tuple(sl)], lr[tuple(sl)]

        def scale(a, clip):
            lo, hi = float(clip[0]), float(clip[1])
            return (np.clip((a - lo) / (hi - lo), 0.0, 1.0) * 2.0 - 1.0).astype(np.float32)
        return scale(hr, self.hr_clip), scale(lr, self.lr_clip)

    def _prepare_all_pairs(self):
        for item in self.pairs:
            sid = item["subject_id"]
            if sid in self.SKIP_SUBJECTS:
                continue
            cache_file = self.cache_dir / f"{sid}_resampled.npz"
            if cache_file.exists():
                with np.load(cache_file) as npz:
                    hr_arr, lr_arr = npz["hr"], npz["lr"]
            else:
                hr_arr, lr_arr = self._load_uncached(item)
                np.savez_compressed(cache_file, hr=hr_arr, lr=lr_arr)
            if hr_arr.ndim != 4 or hr_arr.shape != lr_arr.shape:
                raise ValueError(f"{cache_file}: expected matching [1, H, W, D] arrays, got {hr_arr.shape} / {lr_arr.shape}")
            for s in range(hr_arr.shape[self.slice_axis + 1]):
                self.slice_metadata.append({"hr_arr": hr_arr, "lr_arr": lr_arr, "slice_idx": s, "txt": item["txt"],
                                            "subject_id": sid})

    def __len__(self):
        return len(self.slice_metadata)

    def __getitem__(self, idx):
        m = self.slice_metadata[idx]
        sl = [slice(None)] * 4
        sl[self.slice_axis + 1] = m["slice_idx"]
        hr = torch.from_numpy(np.ascontiguousarray(m["hr_arr"][tuple(sl)][0])).float()
        lr = torch.from_numpy(np.ascontiguousarray(m["lr_arr"][tuple(sl)][0])).float()
        return {"hr": pad_or_center_crop(hr).unsqueeze(0), "lr": pad_or_center_crop(lr).unsqueeze(0), "txt": m["txt"],
                "subject_id": m["subject_id"]}


# ---------------------------------------------------------------------------------------------- DICOM path
def _dicom_reader(path: str) -> np.ndarray:
    try:
        import pydicom
    except ImportError as e:  # pragma: no cover - depends on the image
        raise L.MrisrError("pydicom is not installed: pass slice_reader=... (e.g. numpy.load for .npy slices)") from e
    return pydicom.dcmread(path).pixel_array


class FastMRILazyDataset(Dataset):
    """JSON-indexed lazy slice dataset of nb ResDif c22 (same arguments).  ``__getitem__`` returns the normalised, centre-cropped
    slice ('hr_crop', at most 400 x 400); ``collate`` finishes the batch on the device."""

    CROP = (400, 400)

    def __init__(self, json_path: str, mode: str = "train", target_size: Tuple[int, int] = (512, 512), contrast_filter: str = "T2",
                 strength_filter: str = "3.0T", scale_factor: float = 4.0, fractions: Tuple[float, float, float] = (0.8, 0.1, 0.1),
                 seed: int = 42, slice_reader: Optional[Callable[[str], np.ndarray]] = None, device="cuda"):
        self.target_size, self.scale_factor = tuple(target_size), float(scale_factor)
        self.slice_reader = slice_reader or _dicom_reader
        self.device = torch.device(device)
        with open(json_path, "r") as f:
            self.all_patient_records = json.load(f)
        self.subjects = self._get_filtered_subjects(contrast_filter, strength_filter, seed, fractions, mode)
        self.slice_metadata: List[Dict] = []
        for item in self.subjects:
            for s in self.all_patient_records[item["subject_id"]][item["strength"]][item["contrast"]]:
                self.slice_metadata.append({"path": s["filename"], "subject_id": item["subject_id"], "txt": item["txt"],
                                            "instance": s["instanceNumber"]})

    def _get_filtered_subjects(self, contrast, strength, seed, fractions, mode):
        valid = [{"subject_id": pid, "strength": strength, "contrast": contrast,
                  "txt": f"high quality {contrast} brain MRI, {strength} field strength, medical imaging"}
                 for pid, strengths in self.all_patient_records.items() if strength in strengths and contrast in strengths[strength]]
        train, val, test = random_split(valid, lengths=fractions, generator=torch.Generator().manual_seed(seed))
        sel = {"train": train, "val": val, "test": test}.get(mode, train)
        return [sel.dataset[i] for i in sel.indices]

    def __len__(self):
        return len(self.slice_metadata)

    def __getitem__(self, idx) -> Dict:
        meta = self.slice_metadata[idx]
        arr = np.asarray(self.slice_reader(meta["path"])).astype(np.float32)
        if arr.ndim != 2:
            raise ValueError(f"{meta['path']}: expected a 2-D slice, got shape {arr.shape}")
        lo, hi = arr.min(), arr.max()
        if hi > lo:
            arr = (arr - lo) / (hi - lo)
        h, w = arr.shape
        th, tw = min(h, self.CROP[0]), min(w, self.CROP[1])
        sh, sw = (h - th) // 2, (w - tw) // 2
        crop = np.ascontiguousarray(arr[sh:sh + th, sw:sw + tw])
        return {"hr_crop": torch.from_numpy(crop), "txt": meta["txt"], "subject_id": meta["subject_id"], "instance": meta["instance"]}

    def collate(self, items: Sequence[Dict]) -> Dict:
        """-> {'hr': [B,1,H,W], 'lr': [B,1,H,W]} on the device + the list-valued keys.  ``target_size`` is handed to the resize
        as the reference hands it to PIL, i.e. read as (width, height)."""
        if not items:
            raise ValueError("empty batch")
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise L.MrisrError("slice degradation runs on the GPU (gfx950); there is no CPU fallback")
        out_h, out_w = self.target_size[1], self.target_size[0]
        hr = torch.empty((len(items), out_h, out_w), dtype=torch.float32, device=self.device)
        groups: Dict[Tuple[int, int], List[int]] = {}
        for i, it in enumerate(items):
            groups.setdefault(tuple(it["hr_crop"].shape), []).append(i)
        for shape, idxs in groups.items():  # one resize launch per distinct crop shape (normally just 400 x 400)
            stack = torch.stack([items[i]["hr_crop"] for i in idxs]).to(self.device, non_blocking=True)
            hr[torch.tensor(idxs, device=self.device)] = resize_slices(stack, (out_h, out_w), LANCZOS)
        lr = simulate_low_field(hr, self.scale_factor)
        return {"hr": hr.unsqueeze(1), "lr": lr.unsqueeze(1), "txt": [it["txt"] for it in items],
                "subject_id": [it["subject_id"] for it in items], "instance": [it["instance"] for it in items]}

    def item(self, idx) -> Dict:
        """The reference's ``__getitem__`` result: 'hr' / 'lr' as [1, H, W] tensors (on the device) + scalar metadata."""
        b = self.collate([self[idx]])
        return {"hr": b["hr"][0], "lr": b["lr"][0], "txt": b["txt"][0], "subject_id": b["subject_id"][0], "instance": b["instance"][0]}
