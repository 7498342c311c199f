"""CPU: the metric restatement (oracle/metrics.py) against analytic known answers - torchmetrics / scikit-image are not
installed, so these pins are all there is ("parity unpinned")."""
import numpy as np


def test_psnr_nmse_known_answers():
    from oracle import metrics as om
    gt = np.full((32, 32), 0.5)
    pred = gt + 0.1
    assert abs(om.psnr(pred, gt) - 20.0) < 1e-9          # mse = 0.01 -> 10 log10(1 / 0.01)
    assert abs(om.nmse(pred, gt) - 0.01 / 0.25) < 1e-9    # 0.01 N / (0.25 N)
    assert om.psnr(gt, gt) == float("inf") and om.nmse(gt, gt) == 0.0


def test_ssim_known_answers():
    from oracle import metrics as om
    rng = np.random.default_rng(0)
    x = rng.random((40, 48))
    assert abs(om.ssim(x, x) - 1.0) < 1e-12
    assert abs(om.gaussian_window().sum() - 1.0) < 1e-15 and om.gaussian_window().argmax() == 5
    # constant images: variances and covariance vanish -> SSIM = (2 a b + c1) / (a^2 + b^2 + c1)
    a, b = 0.3, 0.6
    got = om.ssim(np.full((30, 30), a), np.full((30, 30), b))
    assert abs(got - (2 * a * b + 1e-4) / (a * a + b * b + 1e-4)) < 1e-12
    # symmetric in its arguments, and lower for a noisier prediction
    y1, y2 = x + 0.05 * rng.standard_normal(x.shape), x + 0.2 * rng.standard_normal(x.shape)
    assert abs(om.ssim(x, y1) - om.ssim(y1, x)) < 1e-12 and om.ssim(y1, x) > om.ssim(y2, x)


def test_hfen_known_answers():
    from oracle import metrics as om
    rng = np.random.default_rng(1)
    x = rng.random((64, 64))
    assert om.hfen(x, x) == 0.0
    assert abs(om.hfen(2 * x, x) - 1.0) < 1e-6            # LoG is linear: LoG(2x) - LoG(x) = LoG(x)
    assert abs(om.log_filter(np.full((20, 20), 0.7))).max() < 1e-12   # a constant has no high-frequency content
    # a linear ramp is annihilated by the Laplacian away from the borders
    ramp = np.tile(np.arange(64, dtype=np.float64) / 64, (64, 1))
    assert np.abs(om.log_filter(ramp)[10:-10, 10:-10]).max() < 1e-12
