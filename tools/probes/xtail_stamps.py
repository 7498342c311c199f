"""Where the fused transformer-block middle (csrc/xtail.hip) spends its time: clock stamps (s_memtime, 100 MHz) written by wave 0 of
every workgroup at the kernel's phase boundaries, on the bench workload (SD-1.5 + rank-4 LoRA, B = 32, 32 x 32 latents).
Usage (GPU box): python tools/probes/xtail_stamps.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "mri-diffusion-superresolution_amd"))
import mrisr
from mrisr import _lib as L

B = 32
net = mrisr.UNet2DConditionModel(mrisr.UNetConfig(), compute_dtype="bf16", lora_rank=4, lora_alpha=4)
from mrisr import params as P
cfg = mrisr.UNetConfig()
sd = P.random_state_dict(P.unet_param_shapes(cfg), 1, "cuda")
sd.update(P.random_state_dict(P.lora_param_shapes(cfg, 4), 4, "cuda"))
net.load_state_dict(sd)
g = torch.Generator().manual_seed(5)
x = torch.randn((B, 4, 32, 32), generator=g).cuda()
ctx = torch.randn((B, 77, 768), generator=g).cuda()
t = torch.tensor(500).cuda()
for _ in range(3):
    net(x, t, encoder_hidden_states=ctx)
torch.cuda.synchronize()
buf = torch.zeros((256, 40), dtype=torch.int64, device="cuda")
lib = L.lib()
lib.mrisr_debug_xattn_tail_stamps.argtypes = [C.c_void_p]
lib.mrisr_debug_xattn_tail_stamps(C.c_void_p(buf.data_ptr()))
net(x, t, encoder_hidden_states=ctx)
torch.cuda.synchronize()
lib.mrisr_debug_xattn_tail_stamps(C.c_void_p(0))
s = buf.cpu().double()
names = ["start", "rows resident", "z1"] + [f"gemm1 c{c}" for c in range(5)] + ["layernorm", "z2"] + [f"gemm2 c{c}" for c in range(5)] + \
        ["attn heads 0-3", "attn heads 4-7", "z3"] + [f"gemm3 c{c}" for c in range(5)] + ["stores issued", "stores done"]
t0 = s[:, 0].min()
print(f"kernel span (first start -> last end): {(s[:, 24].max() - t0) / 100:.1f} us; workgroup start spread {(s[:, 0].max() - t0) / 100:.1f} us")
for k in range(1, 25):
    d = (s[:, k] - s[:, k - 1]) / 100.0
    print(f"{names[k]:16s} mean {d.mean():7.2f} us  min {d.min():7.2f}  max {d.max():7.2f}   (cumulative mean {((s[:, k] - s[:, 0]) / 100).mean():7.2f})")

print("inside chunk 2 of the second GEMM (wave 0):")
for k, nm in ((26, "vmcnt(0) wait"), (27, "barrier"), (28, "stage + prefetch issue"), (29, "K loop + LoRA step")):
    d = s[:, k] - s[:, k - 1]
    print(f"  {nm:24s} mean {d.mean():8.0f} ticks  min {d.min():8.0f}  max {d.max():8.0f}")
d = s[:, 12] - s[:, 29]
print(f"  {'epilogue + rotate':24s} mean {d.mean():8.0f} ticks  min {d.min():8.0f}  max {d.max():8.0f}")

print("last chunk of the third GEMM (no DMA in flight; wave 0):")
for a_, b_, nm in ((30, 31, "wait + barrier"), (31, 33, "bias copy, first fragment reads"), (33, 34, "K steps 0-4"), (34, 35, "K steps 5-9"), (35, 32, "LoRA step"), (32, 22, "epilogue + rotate")):
    d = s[:, b_] - s[:, a_]
    print(f"  {nm:34s} mean {d.mean():8.0f} ticks  min {d.min():8.0f}  max {d.max():8.0f}")
