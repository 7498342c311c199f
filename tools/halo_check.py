import sys, os
sys.path.insert(0, "/root/repo/mri-diffusion-superresolution_amd")
import torch, torch.nn.functional as F
from mrisr import ops
torch.manual_seed(0)
for (B, C1, C2, Cout, H) in [(4, 320, 0, 320, 32), (4, 320, 320, 320, 32), (4, 640, 0, 640, 16), (2, 64, 0, 64, 32)]:
    x1 = torch.randn(B, C1, H, H).bfloat16().cuda()
    x2 = torch.randn(B, C2, H, H).bfloat16().cuda() if C2 else None
    Cin = C1 + C2
    w = (torch.randn(Cout, Cin, 3, 3) * (9 * Cin) ** -0.5).cuda(); b = torch.randn(Cout).cuda()
    xin = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
    ref = F.conv2d(xin, w.bfloat16().float(), b, padding=1)
    outs = {}
    for t in (25, 14, 41, 42, 43, 44, 45):
        try:
            y = ops.conv3x3(x1, w, b, x2=x2, tile=t, splitk=1).float()
        except Exception as e:
            print("tile", t, "n/a"); continue
        outs[t] = y
        print(f"shape {(B,C1,C2,Cout,H)} tile {t}: rel vs f32 ref {float((y - ref).norm() / ref.norm()):.3e}  max|diff| vs tile25 {float((y - outs[25]).abs().max()):.3e}")
