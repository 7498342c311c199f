"""Oracle: slice degradation of the reference's notebook dataset.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ``FastMRILazyDataset`` in notebooks/ResDif_execution.ipynb code cell 22: ``_center_crop`` (:87-99),
``_pad_to_target`` (:101-113), ``_simulate_low_res`` (:140-154) and the min-max normalisation of ``__getitem__`` (:165-168).
The arithmetic is the reference's own third-party calls - ``scipy.ndimage.gaussian_filter`` and ``PIL.Image.resize`` on
mode "F" images - executed here by the installed libraries (scipy, Pillow 12.2), so this oracle is pinned by running the very
functions the reference runs; only the few lines of glue around them are restated."""
from __future__ import annotations

import numpy as np
from PIL import Image
from scipy.ndimage import gaussian_filter


def center_crop(arr: np.ndarray, crop_size=(400, 400)) -> np.ndarray:
    h, w = arr.shape
    th, tw = min(h, crop_size[0]), min(w, crop_size[1])
    sh, sw = (h - th) // 2, (w - tw) // 2
    return arr[sh:sh + th, sw:sw + tw]


def normalise(arr: np.ndarray) -> np.ndarray:
    arr = arr.astype(np.float32)
    if arr.max() > arr.min():
        arr = (arr - arr.min()) / (arr.max() - arr.min())
    return arr


def pil_resize(arr: np.ndarray, size_wh, resample) -> np.ndarray:
    return np.array(Image.fromarray(np.ascontiguousarray(arr, dtype=np.float32)).resize(tuple(size_wh), resample=resample))


def pad_to_target(arr: np.ndarray, target_size=(512, 512)) -> np.ndarray:
    """center crop 400x400, then LANCZOS resize to ``target_size`` (passed to PIL as given, i.e. read as (width, height))."""
    return pil_resize(center_crop(arr, (400, 400)), target_size, Image.LANCZOS)


def simulate_low_res(hr_arr: np.ndarray, target_size=(512, 512), scale_factor: float = 4.0) -> np.ndarray:
    blurred = gaussian_filter(hr_arr, sigma=0.5 * scale_factor)
    small = (int(target_size[1] // scale_factor), int(target_size[0] // scale_factor))
    lr = Image.fromarray(blurred).resize(small, resample=Image.BICUBIC)
    return np.array(lr.resize(tuple(target_size), resample=Image.BICUBIC))


def reference_item(raw: np.ndarray, target_size=(512, 512), scale_factor: float = 4.0):
    hr = pad_to_target(normalise(raw), target_size)
    return hr, simulate_low_res(hr, target_size, scale_factor)
