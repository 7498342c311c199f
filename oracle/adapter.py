"""Oracle: functional restatement of the reference T2I-Adapter ``Adapter_XL`` as it can actually
run (``sk=True``).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ``/root/reference/src/adapters/modules.py``: ``Downsample`` :52-76, ``ResnetBlock`` :79-111,
``Adapter_XL`` :114-157.  State-dict keys are the reference module's own
(``conv_in.*``, ``body.{k}.{in_conv,block1,block2,down_opt.op}.*``).
Pinned by tests/golden/adapter_xl_tiny.npz (made by importing the reference module).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


@dataclass(frozen=True)
class AdapterConfig:
    channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    nums_rb: int = 3
    cin: int = 192  # 3 * 8 * 8 after PixelUnshuffle(8)
    ksize: int = 3
    use_conv: bool = True  # stride-2 3x3 conv downsample (else 2x2 avg-pool)


ADAPTER_SD15 = AdapterConfig()
ADAPTER_TINY = AdapterConfig(channels=(64, 128, 256, 256), nums_rb=1)


def block_plan(cfg: AdapterConfig) -> List[Tuple[int, int, bool]]:
    """(in_c, out_c, down) of every body block, in order (modules.py:121-134)."""
    plan = []
    for i, c in enumerate(cfg.channels):
        for j in range(cfg.nums_rb):
            if i > 0 and j == 0:
                plan.append((cfg.channels[i - 1], c, True))
            else:
                plan.append((c, c, False))
    return plan


def init_adapter_params(cfg: AdapterConfig = ADAPTER_SD15, seed: int = 20260504) -> Params:
    g = torch.Generator().manual_seed(seed)
    p: Params = {}

    def conv(name, cin, cout, k):
        bound = 1.0 / math.sqrt(cin * k * k)
        p[name + ".weight"] = (torch.rand((cout, cin, k, k), generator=g) * 2 - 1) * bound
        p[name + ".bias"] = (torch.rand((cout,), generator=g) * 2 - 1) * bound

    for k, (ic, oc, down) in enumerate(block_plan(cfg)):
        if ic != oc:  # sk=True: in_conv only where channels change (modules.py:84-87)
            conv(f"body.{k}.in_conv", ic, oc, cfg.ksize)
        conv(f"body.{k}.block1", oc, oc, 3)
        conv(f"body.{k}.block2", oc, oc, cfg.ksize)
        if down and cfg.use_conv:
            conv(f"body.{k}.down_opt.op", ic, ic, 3)  # on the INPUT channels (modules.py:98,69)
    conv("conv_in", cfg.cin, cfg.channels[0], 3)
    return p


def adapter_forward(p: Params, cfg: AdapterConfig, x: torch.Tensor) -> List[torch.Tensor]:
    """[B,3,8h,8w] -> 4 feature maps (modules.py:146-157)."""
    ps = cfg.ksize // 2
    x = F.pixel_unshuffle(x, 8)
    x = F.conv2d(x, p["conv_in.weight"], p["conv_in.bias"], padding=1)
    feats = []
    plan = block_plan(cfg)
    for k, (ic, oc, down) in enumerate(plan):
        b = f"body.{k}"
        if down:
            if cfg.use_conv:
                x = F.conv2d(x, p[b + ".down_opt.op.weight"], p[b + ".down_opt.op.bias"], stride=2, padding=1)
            else:
                x = F.avg_pool2d(x, 2, 2)
        if (b + ".in_conv.weight") in p:
            x = F.conv2d(x, p[b + ".in_conv.weight"], p[b + ".in_conv.bias"], padding=ps)
        h = F.relu(F.conv2d(x, p[b + ".block1.weight"], p[b + ".block1.bias"], padding=1))
        h = F.conv2d(h, p[b + ".block2.weight"], p[b + ".block2.bias"], padding=ps)
        x = h + x
        if (k + 1) % cfg.nums_rb == 0:
            feats.append(x)
    return feats
