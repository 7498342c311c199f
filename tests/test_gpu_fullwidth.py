"""GPU: whole-model parity at FULL SD-1.5 width on the geometry bench.py measures (reference call sites
``src/adapters/res_srdiff.py:65-78``): ``mrisr.UNetConfig()`` = 320/640/1280/1280 channels, head dims 40/80/160, four levels,
rank-4 LoRA on every attention projection, 4x32x32 latents (256^2 px) - and the config-4 shape, ControlNet + UNet at 4x64x64
latents (512^2 px).  Everything the reduced-width tests cover op by op is composed here exactly as the bench composes it: the
shipped tile table (conftest.py points MRISR_TUNE_CACHE at it; its signatures are the B=32 ones), split-K deep convs at 1280
channels, ``attn_fwd_kernel<160,...>``, the 8-byte GroupNorm slabs, the 22-projection time-embedding stack.

Tolerances (written here, as the north star asks): f32 engine vs the CPU oracle 1e-3 relative L2 AND 1e-3 max-relative;
bf16 engine 5e-2 relative L2 against the oracle and against the f32 engine (bf16 storage of every activation, f32 accumulate).
The oracle costs ~0.4 s per sample-step on 16 host threads, so the whole file is about a minute of CPU."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

BF16_TOL = 5e-2


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def maxrel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.fixture(scope="module")
def sd15():
    from oracle import unet as ou
    cfg = ou.SD15
    up = ou.init_unet_params(cfg, seed=1101, perturb_norm=True)
    assert ou.count_params(up) == 859_520_964
    lora = ou.init_lora_params(up, rank=4, seed=1103)
    assert ou.count_params(lora) == 797_184
    return cfg, up, lora


@pytest.fixture(scope="module")
def engines(sd15):
    """One f32 and one bf16 device model of the full network, shared by the tests below."""
    import mrisr
    cfg, up, lora = sd15
    p = {**up, **lora}
    nets = {}
    for dt in ("f32", "bf16"):
        net = mrisr.UNet2DConditionModel(mrisr.UNetConfig(), compute_dtype=dt, lora_rank=4, lora_alpha=4, lora_fused=True,
                                         flash_attention=True)
        net.load_state_dict(p)
        assert net.num_parameters == 859_520_964 + 797_184
        nets[dt] = net
    return nets


_TABLE = os.environ.get("MRISR_TUNE_CACHE", "")
_TABLE_AT_IMPORT = open(_TABLE).read() if _TABLE and os.path.exists(_TABLE) else None


def test_shipped_tile_table_is_in_use():
    assert _TABLE_AT_IMPORT is not None, "conftest.py should point MRISR_TUNE_CACHE at the shipped table"
    assert "32768," in _TABLE_AT_IMPORT  # the B=32 signatures of the bench geometry


def test_sd15_f32_matches_oracle_scalar_and_batched_t(sd15, engines):
    """(i) f32, B=2, 32^2 latents: one scalar-t and one [B]-t forward, 1e-3 rel and max-rel."""
    from oracle import unet as ou
    cfg, up, lora = sd15
    p = {**up, **lora}
    g = torch.Generator().manual_seed(1105)
    x = torch.randn((2, 4, 32, 32), generator=g)
    ctx = torch.randn((2, 77, 768), generator=g)
    for t in (torch.tensor(801), torch.tensor([21, 981])):
        ref = ou.unet_forward(p, cfg, x, t, ctx)
        out = engines["f32"](x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
        e, em = rel(out, ref), maxrel(out, ref)
        print(f"SD-1.5 f32 vs oracle, t={t.tolist()}: rel {e:.3e} maxrel {em:.3e}")
        assert out.shape == ref.shape and e < 1e-3 and em < 1e-3, (e, em)
        ob = engines["bf16"](x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
        eb = rel(ob, ref)
        print(f"SD-1.5 bf16 (B=2, online-tuned tiles) vs oracle: rel {eb:.3e}")
        assert eb < BF16_TOL, eb


def test_sd15_bench_batch_bf16_with_shipped_table(sd15, engines):
    """(ii) B=32 - the M of every signature in the shipped table, i.e. exactly the kernels bench.py times: bf16 against the f32
    engine on the whole batch, and the first two samples of both against the oracle (samples are independent, so rows 0-1 of
    the B=32 result must equal a B=2 forward)."""
    from oracle import unet as ou
    cfg, up, lora = sd15
    p = {**up, **lora}
    g = torch.Generator().manual_seed(1107)
    x = torch.randn((32, 4, 32, 32), generator=g)
    ctx = torch.randn((32, 77, 768), generator=g)
    t = torch.tensor(641)
    f32 = engines["f32"](x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
    b16 = engines["bf16"](x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
    ref = ou.unet_forward(p, cfg, x[:2], t, ctx[:2])
    e32, e16, ed = rel(f32[:2], ref), rel(b16[:2], ref), rel(b16, f32)
    print(f"SD-1.5 B=32: f32[:2] vs oracle {e32:.3e} (max {maxrel(f32[:2], ref):.3e}); bf16[:2] vs oracle {e16:.3e}; bf16 vs f32 engine {ed:.3e}")
    assert e32 < 1e-3 and maxrel(f32[:2], ref) < 1e-3
    assert e16 < BF16_TOL and ed < BF16_TOL
    # per-sample: no sample of the batch is off (a wrong tile at one M offset would hide in the batch norm)
    per = ((b16.float() - f32.float()).flatten(1).norm(dim=1) / f32.float().flatten(1).norm(dim=1)).cpu()
    assert float(per.max()) < BF16_TOL, per


def test_sd15_consumes_the_reference_producers_fixed_embeds(sd15, engines, golden_dir):
    """SURVEY.md 8 a10 at the real width: the `[1, 77, 768]` tensor the REFERENCE's `get_fixed_prompt_embeds` returned for the
    stub tokenizer / encoder (tests/golden/prompt_embeds.npz, made by tests/golden/make_golden.py) as fp16 on the device,
    sliced `[0:1]` as `log_validation` slices it (res_srdiff.py:67,75), B = 1."""
    import numpy as np
    from oracle import unet as ou
    cfg, up, lora = sd15
    fixed = torch.from_numpy(np.load(os.path.join(golden_dir, "prompt_embeds.npz"))["fixed"])
    assert tuple(fixed.shape) == (1, 77, 768)
    f16 = fixed.to(torch.float16).cuda()
    g = torch.Generator().manual_seed(1119)
    x = torch.randn((1, 4, 32, 32), generator=g)
    ref = ou.unet_forward({**up, **lora}, cfg, x, torch.tensor(961), f16.float().cpu())
    out = engines["f32"](x.cuda(), torch.tensor(961).cuda(), encoder_hidden_states=f16[0:1]).sample
    assert rel(out, ref) < 1e-3 and maxrel(out, ref) < 1e-3
    assert rel(engines["bf16"](x.cuda(), torch.tensor(961).cuda(), encoder_hidden_states=f16[0:1]).sample, ref) < BF16_TOL


def test_sd15_bf16_forward_with_poisoned_lds_gives_the_same_bits(engines):
    """The whole SD-1.5 forward at the bench batch (the shipped table's kernels, two workgroups per CU everywhere) with every
    GEMM workgroup's LDS pre-filled with NaN bytes (`mrisr_debug_gemm_flags(2048)`, see test_gpu_ops.py): bit-identical to the
    normal run - no kernel of the step reads LDS that its own DMA has not written."""
    import ctypes as C
    from mrisr import _lib as L
    g = torch.Generator().manual_seed(1117)
    x = torch.randn((32, 4, 32, 32), generator=g).cuda()
    ctx = torch.randn((32, 77, 768), generator=g).cuda()
    t = torch.tensor(333).cuda()
    clean = engines["bf16"](x, t, encoder_hidden_states=ctx).sample.clone()
    try:
        L.lib().mrisr_debug_gemm_flags(C.c_int(2048))
        for rep in range(2):
            got = engines["bf16"](x, t, encoder_hidden_states=ctx).sample
            assert bool(torch.isfinite(got.float()).all()), rep
            assert torch.equal(got, clean), (rep, float((got.float() - clean.float()).abs().max()))
    finally:
        L.lib().mrisr_debug_gemm_flags(C.c_int(0))


def test_sd15_ddim_three_steps_through_the_captured_graph(sd15, engines):
    """(iii) three DDIM steps through the hipGraph-captured sampler vs oracle.sampler.ddim_sample: f32 at B=2 (1e-3), bf16 at the
    bench batch B=32 (rows 0-1 against the same oracle trajectory)."""
    import mrisr
    from oracle import sampler as osa
    from oracle import schedulers as osch
    from oracle import unet as ou
    cfg, up, lora = sd15
    p = {**up, **lora}
    g = torch.Generator().manual_seed(1109)
    x = torch.randn((32, 4, 32, 32), generator=g)
    ctx = torch.randn((32, 77, 768), generator=g)
    so = osch.OracleScheduler(timestep_spacing="leading", steps_offset=1)
    so.set_timesteps(50)
    so.timesteps = so.timesteps[:3]  # the first three steps of the 50-step schedule (t = 981, 961, 941)
    traj = osa.ddim_sample(ou.OracleUNet(p, cfg), x[:2], ctx[:2], so)
    sp = mrisr.DDIMScheduler(timestep_spacing="leading", steps_offset=1)
    sp.set_timesteps(50)
    lat = x[:2].cuda().clone().contiguous()
    smp = mrisr.Sampler(engines["f32"], sp, kind="ddim")
    smp.set_range(0, 3)
    smp.run(lat, ctx[:2].cuda(), use_graph=True)
    torch.cuda.synchronize()
    e = rel(lat, traj[-1])
    print(f"SD-1.5 f32 3 DDIM steps (graph) vs oracle: rel {e:.3e} maxrel {maxrel(lat, traj[-1]):.3e}")
    assert e < 1e-3 and maxrel(lat, traj[-1]) < 1e-3
    for use_graph in (True, False):
        lat = x.cuda().clone().contiguous()
        smp = mrisr.Sampler(engines["bf16"], sp, kind="ddim")
        smp.set_range(0, 3)
        smp.run(lat, ctx.cuda(), use_graph=use_graph)
        torch.cuda.synchronize()
        eb = rel(lat[:2], traj[-1])
        print(f"SD-1.5 bf16 B=32 3 DDIM steps (graph={use_graph}) rows 0-1 vs oracle: rel {eb:.3e}")
        assert torch.isfinite(lat).all() and eb < BF16_TOL, eb


@pytest.mark.parametrize("dt,tol", [("f32", 1e-3), ("bf16", BF16_TOL)])
def test_config4_controlnet_plus_unet_at_64x64(sd15, engines, dt, tol):
    """(iv) BASELINE config 4 shape: full ControlNet encoder + UNet at 4x64x64 latents (512^2 px), B=2, one step - the
    ControlNet's 12 + 1 residuals and the UNet fed with them, against the oracle."""
    import mrisr
    from oracle import unet as ou
    cfg, up, lora = sd15
    p = {**up, **lora}
    cp = ou.init_controlnet_params(cfg, seed=1111, perturb_norm=True)
    assert ou.count_params(cp) == 361_279_120
    g = torch.Generator().manual_seed(1113)
    x = torch.randn((2, 4, 64, 64), generator=g)
    ctx = torch.randn((2, 77, 768), generator=g)
    cond = torch.randn((2, 3, 512, 512), generator=g)
    t = torch.tensor(501)
    dref, mref = ou.controlnet_forward(cp, cfg, x, t, ctx, cond)
    ref = ou.unet_forward(p, cfg, x, t, ctx, dref, mref)
    cnet = mrisr.ControlNetModel(mrisr.UNetConfig(), compute_dtype=dt)
    cnet.load_state_dict(cp)
    down, mid = cnet(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda(), controlnet_cond=cond.cuda(), return_dict=False)
    errs = [rel(a, b) for a, b in zip(list(down) + [mid], dref + [mref])]
    print(f"config 4 [{dt}] ControlNet residuals vs oracle: max rel {max(errs):.3e}")
    assert len(down) == 12 and max(errs) < tol, errs
    out = engines[dt](x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda(), down_block_additional_residuals=list(down),
                      mid_block_additional_residual=mid).sample
    e = rel(out, ref)
    print(f"config 4 [{dt}] UNet(ControlNet residuals) at 64x64 vs oracle: rel {e:.3e}")
    assert e < tol, e
    if dt == "f32":
        assert maxrel(out, ref) < 1e-3


def test_tile_table_not_modified_by_the_run():
    """The shipped table is read-only for test / bench runs (MRISR_TUNE_WRITE unset): the B=2 and 64x64 signatures tuned online
    by the tests above stay in memory instead of being appended to a tracked file."""
    assert open(_TABLE).read() == _TABLE_AT_IMPORT
