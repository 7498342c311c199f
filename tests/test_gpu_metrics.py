"""GPU: PSNR / SSIM / HFEN / NMSE kernels of libmrisr against the CPU restatement of the reference's evaluator."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _pair(seed, H, W):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    gt = 0.5 + 0.4 * np.sin(xx / 7.0) * np.cos(yy / 5.0)
    gt = np.clip(gt + 0.05 * rng.standard_normal((H, W)), 0, 1)
    pred = np.clip(gt + 0.03 * rng.standard_normal((H, W)) + 0.02, 0, 1)
    q = lambda a: np.round(a * 255).astype(np.uint8)  # noqa: E731  8-bit, like the PNGs the reference reads
    return q(pred), q(gt)


@pytest.mark.parametrize("H,W", [(256, 256), (64, 96), (33, 47), (512, 512)])
def test_metrics_match_oracle(H, W):
    import mrisr
    from oracle import metrics as om
    ev = mrisr.MRIEvaluator()
    ps, gs = zip(*[_pair(s, H, W) for s in range(3)])
    p = np.stack(ps).astype(np.float32) / 255.0
    g = np.stack(gs).astype(np.float32) / 255.0
    r = ev.evaluate(torch.from_numpy(p)[:, None], torch.from_numpy(g)[:, None])
    for i in range(3):
        ref = om.evaluate(p[i], g[i])
        assert abs(float(r["PSNR"][i]) - ref["PSNR"]) < 1e-3, (float(r["PSNR"][i]), ref["PSNR"])   # "identical to 3 s.f."
        assert abs(float(r["SSIM"][i]) - ref["SSIM"]) < 1e-4
        assert abs(float(r["HFEN"][i]) - ref["HFEN"]) < 1e-4 * max(1.0, ref["HFEN"])
        assert abs(float(r["NMSE"][i]) - ref["NMSE"]) < 1e-6
    # the reference's call surface (single image, torchmetrics-like callables)
    one_p, one_g = torch.from_numpy(p[:1])[:, None].cuda(), torch.from_numpy(g[:1])[:, None].cuda()
    ref = om.evaluate(p[0], g[0])
    assert abs(ev.psnr(one_p, one_g).item() - ref["PSNR"]) < 1e-3
    assert abs(ev.ssim(one_p, one_g).item() - ref["SSIM"]) < 1e-4
    assert abs(ev.compute_hfen(p[0], g[0]) - ref["HFEN"]) < 1e-4 * max(1.0, ref["HFEN"])
    assert abs(ev.compute_nmse(p[0], g[0]) - ref["NMSE"]) < 1e-6


def test_identical_images_and_errors():
    import mrisr
    ev = mrisr.MRIEvaluator()
    x = torch.rand((2, 1, 40, 40))
    r = ev.evaluate(x, x)
    assert torch.isinf(r["PSNR"]).all() and (r["SSIM"] - 1).abs().max() < 1e-6 and (r["HFEN"] == 0).all() and (r["NMSE"] == 0).all()
    with pytest.raises(ValueError, match="differ"):
        ev.evaluate(x, x[:, :, :30])
    with pytest.raises(ValueError, match="larger than"):
        ev.evaluate(torch.rand(8, 8), torch.rand(8, 8))


def test_evaluate_folders_contract(tmp_path):
    """Two folders of same-named grayscale PNGs (the reference's input contract, eval.py:53-116)."""
    from PIL import Image
    import mrisr
    from oracle import metrics as om
    gd, td = tmp_path / "gen", tmp_path / "gt"
    gd.mkdir(); td.mkdir()
    refs = []
    for i in range(4):
        p, g = _pair(10 + i, 64, 64)
        Image.fromarray(p).save(gd / f"s{i:03d}.png")
        Image.fromarray(g).save(td / f"s{i:03d}.png")
        refs.append(om.evaluate(p.astype(np.float32) / 255, g.astype(np.float32) / 255))
    ev = mrisr.MRIEvaluator()
    out = ev.evaluate_folders(str(gd), str(td))
    for k in ("PSNR", "SSIM", "HFEN", "NMSE"):
        assert abs(out[k] - np.mean([r[k] for r in refs])) < 1e-3
    bug = ev.evaluate_folders(str(gd), str(td), reference_count_bug=True)   # eval.py:91 `count += 13`
    assert abs(bug["PSNR"] * 13 - out["PSNR"]) < 1e-3
    assert ev.evaluate_folders(str(tmp_path / "gen"), str(tmp_path)) is None
