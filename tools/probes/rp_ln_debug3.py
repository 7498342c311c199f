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
flags = int(os.environ.get("MRISR_GEMM_FLAGS", "0"))
xin = x.float() if flags & (2048 | 4096) else F.layer_norm(x.float(), (K,), ga, be, 1e-5).to(torch.bfloat16).float()
ref = F.linear(xin, w.to(torch.bfloat16).float(), b)
got = ops.ln_linear(x.cuda(), ga.cuda(), be.cuda(), w.cuda(), b.cuda()).float().cpu()
err = (got - ref).abs() > 0.05 * ref.abs().max()
print("flags", flags, "rel", float((got - ref).norm() / ref.norm()), "nbad", int(err.sum()), "bad rows mod 32:", sorted(set((err.nonzero()[:, 0] % 32).tolist())))
