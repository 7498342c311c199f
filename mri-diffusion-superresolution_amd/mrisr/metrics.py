"""``MRIEvaluator`` - host mirror of the reference's ``src/eval/eval.py`` (same method names and folder contract); the
PSNR / SSIM / HFEN / NMSE arithmetic runs in ``libmrisr.so`` (``csrc/metrics.hip``).  No CPU fallback.

Differences worth knowing: images are read with PIL (``convert("L")``) instead of ``cv2.imread(..., IMREAD_GRAYSCALE)`` -
identical for the grayscale PNGs the sampler writes; and the reference's averaging bug (``count += 13`` per image,
eval.py:91, so every mean is 13x too small) is reproduced only when ``reference_count_bug=True`` is passed."""
from __future__ import annotations

import ctypes as C
import glob
import os
from typing import Dict

import numpy as np
import torch

from . import _lib as L


class MRIEvaluator:
    def __init__(self, device="cuda"):
        if not torch.cuda.is_available():
            raise L.MrisrError("mrisr needs an AMD GPU (gfx950); there is no CPU fallback")
        self.device = torch.device(device)

    # ---- core: per-image metrics of a batch ----
    def evaluate(self, pred, target) -> Dict[str, torch.Tensor]:
        """pred / target: [B,1,H,W], [B,H,W] or [H,W], values in [0,1] -> dict of [B] tensors."""
        p, t = self._prep(pred), self._prep(target)
        if p.shape != t.shape:
            raise ValueError(f"pred and target shapes differ: {tuple(p.shape)} vs {tuple(t.shape)}")
        B, H, W = p.shape
        if H <= 10 or W <= 10:
            raise ValueError("images must be larger than the 11x11 SSIM window")
        scratch = torch.empty((2, B, H, W), dtype=torch.float32, device=self.device)
        sums = torch.empty((B, 6), dtype=torch.float64, device=self.device)
        out = torch.empty((B, 4), dtype=torch.float32, device=self.device)
        L.check(L.lib().mrisr_image_metrics(C.c_void_p(p.data_ptr()), C.c_void_p(t.data_ptr()), B, H, W, C.c_void_p(scratch.data_ptr()),
                                            C.c_void_p(sums.data_ptr()), C.c_void_p(out.data_ptr()), L.stream_ptr()))
        self._sums = sums
        return {"PSNR": out[:, 0], "SSIM": out[:, 1], "HFEN": out[:, 2], "NMSE": out[:, 3]}

    def _prep(self, x) -> torch.Tensor:
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        x = x.to(self.device, torch.float32)
        if x.ndim == 4:
            if x.shape[1] != 1:
                raise ValueError("metrics are defined on single-channel images [B,1,H,W]")
            x = x[:, 0]
        elif x.ndim == 2:
            x = x[None]
        elif x.ndim != 3:
            raise ValueError(f"expected [B,1,H,W], [B,H,W] or [H,W]; got {tuple(x.shape)}")
        return x.contiguous()

    # ---- the reference's attribute surface ----
    def psnr(self, pred, target) -> torch.Tensor:
        """torchmetrics PeakSignalNoiseRatio(data_range=1.0): one value over everything passed in."""
        r = self.evaluate(pred, target)
        if r["PSNR"].numel() == 1:
            return r["PSNR"][0]
        p = self._prep(pred)
        mse = self._sums[:, 0].sum() / p.numel()
        return (10.0 * torch.log10(1.0 / mse)).float()

    def ssim(self, pred, target) -> torch.Tensor:
        return self.evaluate(pred, target)["SSIM"].mean()

    def compute_hfen(self, pred, target, sigma: float = 1.5) -> float:
        if sigma != 1.5:
            raise ValueError("the device HFEN is built for the reference's sigma = 1.5")
        return float(self.evaluate(pred, target)["HFEN"].mean())

    def compute_nmse(self, pred, target) -> float:
        return float(self.evaluate(pred, target)["NMSE"].mean())

    def evaluate_folders(self, generated_dir, ground_truth_dir, reference_count_bug: bool = False):
        from PIL import Image
        exts = ["*.png", "*.jpg", "*.JPG"]
        gen = sorted(f for e in exts for f in glob.glob(os.path.join(generated_dir, e)))
        gt = sorted(f for e in exts for f in glob.glob(os.path.join(ground_truth_dir, e)))
        if len(gen) != len(gt):
            print(f"Warning: File count mismatch. Gen: {len(gen)}, GT: {len(gt)}")
        sums = {"PSNR": 0.0, "SSIM": 0.0, "HFEN": 0.0, "NMSE": 0.0}
        count = 0
        for a, b in zip(gen, gt):
            try:
                ia = np.asarray(Image.open(a).convert("L"), dtype=np.float32) / 255.0
                ib = np.asarray(Image.open(b).convert("L"), dtype=np.float32) / 255.0
            except OSError:
                print(f"Error reading pair: {a}")
                continue
            r = self.evaluate(ia, ib)
            for k in sums:
                sums[k] += float(r[k][0])
            count += 13 if reference_count_bug else 1
        if count == 0:
            print("No images processed.")
            return None
        return {k: v / count for k, v in sums.items()}
