import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch, torch.nn.functional as F
from mrisr import ops
def sdpa(q, k, v, H):
    B, N, C = q.shape; d = C // H
    qh, kh, vh = (t.float().view(B, -1, H, d).transpose(1, 2) for t in (q, k, v))
    return F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, N, C)
torch.manual_seed(0)
for (B, N, Nk, C, H) in [(2, 256, 256, 320, 8), (1, 1024, 1024, 320, 8), (2, 64, 64, 1280, 8), (2, 16, 16, 1280, 8), (2, 256, 77, 640, 8), (1, 200, 77, 64, 8), (1, 64, 64, 256, 8)]:
    q, k, v = (torch.randn(B, n, C).bfloat16() for n in (N, Nk, Nk))
    ref = sdpa(q, k, v, H)
    y = ops.attention(q.cuda(), k.cuda(), v.cuda(), H, flash=True).float().cpu()
    e = float((y - ref).norm() / ref.norm())
    ratio = (y / ref.clamp_min(1e-3).where(ref.abs() > 1e-2, torch.ones_like(ref)))
    print(f"B={B} N={N} Nk={Nk} C={C} hd={C//H}: rel {e:.4f}  nan={int(torch.isnan(y).sum())}  mean|y|/mean|ref| = {float(y.abs().mean() / ref.abs().mean()):.4f}")
