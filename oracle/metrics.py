"""ORACLE (test infrastructure only): CPU restatement of the reference's image metrics, ``src/eval/eval.py``:

    PSNR  torchmetrics PeakSignalNoiseRatio(data_range=1.0)                      eval.py:15, 85
    SSIM  torchmetrics StructuralSimilarityIndexMeasure(data_range=1.0)          eval.py:16, 86
    HFEN  ||LoG(pred) - LoG(gt)|| / (||LoG(gt)|| + 1e-8), LoG = laplace(gaussian(., 1.5))   eval.py:18-37
    NMSE  ||pred - gt||^2 / (||gt||^2 + 1e-8)                                    eval.py:39-51

torchmetrics and scikit-image are not installed here, so their arithmetic is restated from their published
definitions and pinned only by analytic known answers (tests/test_oracle_metrics.py) - "parity unpinned":

* SSIM (torchmetrics defaults): 11x11 Gaussian window, sigma 1.5, k1 = 0.01, k2 = 0.03; the image is reflect-padded by 5,
  filtered, and the 5-pixel border is cropped again before averaging, i.e. the mean runs over the windows that lie fully
  inside the image.
* skimage.filters.gaussian(sigma): scipy.ndimage.gaussian_filter(mode="nearest", truncate=4.0) -> radius int(4*sigma + 0.5).
* skimage.filters.laplace(ksize=3): scipy.ndimage.convolve(image, [[0,-1,0],[-1,4,-1],[0,-1,0]], mode="reflect").
Images: float in [0, 1] (the reference divides the 8-bit PNG by 255), shape [H, W].
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage


def psnr(pred: np.ndarray, gt: np.ndarray, data_range: float = 1.0) -> float:
    mse = float(np.mean((pred.astype(np.float64) - gt.astype(np.float64)) ** 2))
    return float(10.0 * np.log10(data_range ** 2 / mse)) if mse > 0 else float("inf")


def gaussian_window(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    x = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = np.exp(-(x / sigma) ** 2 / 2.0)
    return g / g.sum()


def ssim(pred: np.ndarray, gt: np.ndarray, data_range: float = 1.0, size: int = 11, sigma: float = 1.5, k1: float = 0.01,
         k2: float = 0.03) -> float:
    x, y = pred.astype(np.float64), gt.astype(np.float64)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    g = gaussian_window(size, sigma)
    pad = (size - 1) // 2

    def filt(a):  # separable window over the fully-inside positions only
        a = np.apply_along_axis(lambda r: np.convolve(r, g, mode="valid"), 1, a)
        return np.apply_along_axis(lambda c: np.convolve(c, g, mode="valid"), 0, a)

    assert x.shape[0] > 2 * pad and x.shape[1] > 2 * pad, "image smaller than the SSIM window"
    mx, my = filt(x), filt(y)
    sxx, syy, sxy = filt(x * x) - mx * mx, filt(y * y) - my * my, filt(x * y) - mx * my
    s = ((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2))
    return float(s.mean())


def log_filter(img: np.ndarray, sigma: float = 1.5) -> np.ndarray:
    sm = ndimage.gaussian_filter(img.astype(np.float64), sigma=sigma, mode="nearest", truncate=4.0)
    lap = np.array([[0.0, -1.0, 0.0], [-1.0, 4.0, -1.0], [0.0, -1.0, 0.0]])
    return ndimage.convolve(sm, lap, mode="reflect")


def hfen(pred: np.ndarray, gt: np.ndarray, sigma: float = 1.5) -> float:
    a, b = log_filter(pred, sigma), log_filter(gt, sigma)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-8))


def nmse(pred: np.ndarray, gt: np.ndarray) -> float:
    p, t = pred.astype(np.float64), gt.astype(np.float64)
    return float(np.linalg.norm(p - t) ** 2 / (np.linalg.norm(t) ** 2 + 1e-8))


def evaluate(pred: np.ndarray, gt: np.ndarray) -> dict:
    return {"PSNR": psnr(pred, gt), "SSIM": ssim(pred, gt), "HFEN": hfen(pred, gt), "NMSE": nmse(pred, gt)}
