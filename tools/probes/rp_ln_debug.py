import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch, torch.nn.functional as F
from mrisr import ops
torch.manual_seed(0)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 960)]
for M, N in shapes:
    K = 320
    x = (torch.randn(M, K) * 0.5 + 3.0).to(torch.bfloat16)
    w = torch.randn(N, K) * K ** -0.5
    b = torch.randn(N)
    ga, be = 1 + 0.1 * torch.randn(K), 0.1 * torch.randn(K)
    xn = F.layer_norm(x.float(), (K,), ga, be, 1e-5).to(torch.bfloat16).float()
    ref = F.linear(xn, w.to(torch.bfloat16).float(), b)
    ref0 = F.linear(x.float(), w.to(torch.bfloat16).float(), b)
    for rep in range(2):
        got = ops.ln_linear(x.cuda(), ga.cuda(), be.cuda(), w.cuda(), b.cuda()).float().cpu()
        got0 = ops.linear(x.cuda(), w.cuda(), b.cuda(), tile=60).float().cpu()
        err = (got - ref)
        e = float(err.norm() / ref.norm())
        e0 = float((got0 - ref0).norm() / ref0.norm())
        bad = (err.abs() > 0.05 * ref.abs().max())
        rows_in_panel = torch.zeros(128); cols_in_chunk = torch.zeros(64)
        idx = bad.nonzero()
        for r, c in idx[:20000].tolist():
            rows_in_panel[r % 128] += 1; cols_in_chunk[c % 64] += 1
        print(f"M={M} N={N} ysplit={os.environ.get('MRISR_RP_YSPLIT','auto')} rep={rep} LN rel={e:.4f} plain rel={e0:.4f} nbad={int(bad.sum())} "
              f"rows%128 nonzero={int((rows_in_panel>0).sum())} {rows_in_panel[:8].tolist()} cols%64 nonzero={int((cols_in_chunk>0).sum())} {cols_in_chunk[:8].tolist()}")
