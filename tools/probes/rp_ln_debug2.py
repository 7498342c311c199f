import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch, torch.nn.functional as F
from mrisr import ops
torch.manual_seed(0)
M, N, K = 4096, 960, 320
x = (torch.randn(M, K) * 0.5 + 3.0).to(torch.bfloat16)
w = torch.randn(N, K) * K ** -0.5
b = torch.randn(N)
ga, be = torch.ones(K), torch.zeros(K)
xn = F.layer_norm(x.float(), (K,), ga, be, 1e-5).to(torch.bfloat16).float()
ref = F.linear(xn, w.to(torch.bfloat16).float(), b)
got = ops.ln_linear(x.cuda(), ga.cuda(), be.cuda(), w.cuda(), b.cuda()).float().cpu()
err = (got - ref).abs() > 0.05 * ref.abs().max()
rows = sorted(set((err.nonzero()[:, 0] % 128).tolist()))
print("bad rows mod 128:", rows)
cols = sorted(set((err.nonzero()[:, 1] % 64).tolist()))
print("bad cols mod 64:", cols)
r, c = err.nonzero()[0].tolist()
print("example", r, c, float(got[r, c]), float(ref[r, c]))
# is the bad value consistent with an un-normalised row (plain x W^T + b)?
plain = F.linear(x.float(), w.to(torch.bfloat16).float(), b)
print("bad entries close to plain(x)?", float((got[err] - plain[err]).abs().mean()), "vs ref", float((got[err] - ref[err]).abs().mean()))
# per row: is the whole row bad or some columns?
rb = err[:, 512:].float().mean(1)
print("fraction of bad cols (cols>=512) in bad rows: min/mean/max", float(rb[rb > 0].min()), float(rb[rb > 0].mean()), float(rb.max()), "rows with any bad:", int((rb > 0).sum()))
