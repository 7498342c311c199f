import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch
import mrisr
from mrisr import _lib as L
from oracle import unet as ou
torch.set_grad_enabled(False)
cfg = ou.UNetConfig(block_out_channels=(320, 640), attn_levels=(True, True), cross_attention_dim=64)
up = ou.init_unet_params(cfg, seed=71, perturb_norm=True)
lora = ou.init_lora_params(up, rank=4, seed=72)
g = torch.Generator().manual_seed(73)
x = torch.randn((2, 4, 16, 16), generator=g); ctx = torch.randn((2, 77, 64), generator=g); t = torch.tensor([40, 700])
ref = ou.unet_forward({**up, **lora}, cfg, x, t, ctx)
ref_nolora = ou.unet_forward(up, cfg, x, t, ctx)
def rel(a, b): return float((a.float().cpu() - b).norm() / b.norm())
lib = L.lib()
for pref in (0, 50, 43):
    lib.mrisr_debug_prefer_tile(C.c_int(pref))
    for fused in (True, False):
        net = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4, lora_fused=fused)
        net.load_state_dict({**up, **lora})
        print("pref", pref, "lora_fused", fused, "rel", rel(net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample, ref))
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16")
    net.load_state_dict(up)
    print("pref", pref, "no lora rel", rel(net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample, ref_nolora))
net = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4)
net.load_state_dict({**up, **lora})
print("f32 rel", rel(net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample, ref))
