#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): MRI slices/sec for a 50-step DDIM(eta=0) sample at 256^2 px
(-> 4x32x32 latents), B=32 slices per GPU, SD-1.5-size UNet + rank-4 LoRA fused into the projection GEMMs.

One "step" = one batch of 32 synthetic slices through all 50 denoising steps (50 hipGraph replays of
UNet forward + fused DDIM update), inputs resident in HBM.  One process per GPU; inference shards over
independent slices, so there is NO data-path collective (weak scaling) - torch.distributed is used only for
the barrier and the max-over-ranks time.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...          (no launcher: starts its own N children, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for how every field is derived).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# Tile / split-K table of the bench geometry, measured on an MI355X by the plan-time autotuner (gemm_choose) and shipped
# like a GEMM library's logic file: the tuner's 3-launch timings are noisy enough to move the end-to-end figure by +-2 %
# from run to run, a fixed table makes it repeatable.  Shapes that are not in the table are still tuned online (and
# appended).  Override with MRISR_TUNE_CACHE=<path>, or MRISR_TUNE_CACHE= (empty) for pure online tuning.
os.environ.setdefault("MRISR_TUNE_CACHE", os.path.join(ROOT, "profiles", "r03_tune_cache.tsv"))
if not os.environ["MRISR_TUNE_CACHE"]:
    del os.environ["MRISR_TUNE_CACHE"]
for p in (ROOT, os.path.join(ROOT, "mri-diffusion-superresolution_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

SEED = 20260501
B_PER_GPU = 32
LATENT = 32
N_DDIM = 50
UNET_GFLOP_PER_SAMPLE = 180.27  # algorithmic 2*MAC FLOPs of one UNet forward @32^2 latents
PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


T_START = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_threads():
    """CPU share of this process (cgroup/affinity aware), capped at 16 = one GPU's share of the box."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def synthetic_batch(B, device, rank):
    """Procedural brain-like phantoms -> LR anchor latents (stub VAE: avgpool8 * 0.18215), context, x_T."""
    g = torch.Generator(device=device).manual_seed(SEED + 1000 * rank)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 256, device=device), torch.linspace(-1, 1, 256, device=device), indexing="ij")
    imgs = []
    for i in range(B):
        r = torch.rand(6, device=device, generator=g)
        head = ((xx / (0.7 + 0.1 * r[0])) ** 2 + (yy / (0.85 + 0.1 * r[1])) ** 2 < 1).float()
        inner = ((xx - 0.2 * (r[2] - 0.5)) ** 2 / 0.25 + (yy - 0.2 * (r[3] - 0.5)) ** 2 / 0.36 < 1).float()
        tex = torch.nn.functional.interpolate(torch.randn(1, 1, 16, 16, device=device, generator=g), size=(256, 256),
                                              mode="bicubic", align_corners=False)[0, 0]
        imgs.append((0.6 * head + 0.3 * inner + 0.1 * tex * head).clamp(0, 1) * 2 - 1)
    hr = torch.stack(imgs)[:, None]
    lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 4), scale_factor=4.0, mode="bicubic").clamp(-1, 1)
    lr_lat = (torch.nn.functional.avg_pool2d(lr, 8).repeat(1, 4, 1, 1) * 0.18215).contiguous()
    ctx = torch.randn((B, 77, 768), device=device, generator=g)
    noise = torch.randn(lr_lat.shape, device=device, generator=g)
    return lr_lat, ctx, noise, hr


def stub_decode(z):
    """Stub VAE decoder of SURVEY.md 8d (the real AutoencoderKL is a separate module, not part of the timed path): latents ->
    [0, 1] grey slice at 8x the size, the way the reference post-processes a decoded sample (res_srdiff.py:113)."""
    img = torch.nn.functional.interpolate((z / 0.18215).mean(1, keepdim=True), scale_factor=8.0, mode="nearest")
    return (img / 2 + 0.5).clamp(0, 1)


def psnr(a, b):
    """torchmetrics' PeakSignalNoiseRatio(data_range=1.0) as src/eval/eval.py:15 uses it: 10 log10(1 / mse)."""
    mse = float(((a.double() - b.double()) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * __import__("math").log10(1.0 / mse)


def prof_report(lib):
    buf = C.create_string_buffer(1 << 20)
    n = lib.mrisr_prof_report(buf, len(buf))
    return json.loads(buf.value.decode()) if n > 0 else {}


def cpu_baseline(state_dict_cpu, cfg_oracle, threads, x_T, ctx, n_steps):
    """The oracle (CPU restatement of the reference's diffusers path) on this box's host cores: the WHOLE 50-step DDIM loop
    of ONE slice of the workload - the first slice of rank 0's batch, same x_T and context as the GPU run - so the same
    bounded sample (about 20-30 s on 16 threads) gives the CPU rate (not extrapolated) and the reference result the GPU
    outputs are scored against (the "PSNR vs ref" half of the metric)."""
    from oracle import sampler as osa
    from oracle import schedulers as osch
    from oracle import unet as ou
    torch.set_num_threads(threads)
    so = osch.OracleScheduler(timestep_spacing="leading", steps_offset=1)
    so.set_timesteps(n_steps)
    with torch.no_grad():
        t0 = time.perf_counter()
        traj = osa.ddim_sample(ou.OracleUNet(state_dict_cpu, cfg_oracle), x_T, ctx, so)
        dt = time.perf_counter() - t0
    B = x_T.shape[0]
    return {"value": B / dt, "unit": "slices/s", "cores": threads, "kind": "port",
            "sample": f"oracle (CPU restatement of the diffusers path, fp32) {n_steps}-step DDIM of {B} slice of the workload, "
                      f"whole loop, {dt:.1f} s; not extrapolated",
            "sample_steps_per_s": B * n_steps / dt}, traj[-1]


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): this process has made NO GPU call (importing torch
    does not initialise HIP), so it may start N fresh children - one per GPU, the same environment torchrun would give them
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) - relay their output (only rank 0 prints the JSON line) and
    exit with the worst child's code.  Children are separate processes (no exec from a GPU-initialised process anywhere)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        log(f"self-launch: ranks failed (rank, rc): {bad}")
    return max(((128 - rc) if rc < 0 else rc) for rc in rcs)  # a child killed by signal s counts as 128 + s


def launcher_selftest(world, rank):
    """`--launcher-selftest` (CPU, gloo): the rendezvous, the barrier and the max-over-ranks reduce of the bench contract
    without a model - what tests/test_dist_cpu.py drives through the self-launcher."""
    import torch.distributed as dist
    from mrisr import dist as mdist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    slowest = mdist.max_over_ranks(1.0 + rank)
    seen = mdist.sum_over_ranks(1.0)
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "ranks_seen": int(seen), "slowest": slowest,
                          "scaling": "weak"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="slices per GPU (BASELINE config: 32)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lora-merged", action="store_true", help="merge LoRA into W instead of the fused rank tail")
    ap.add_argument("--ddim-steps", type=int, default=N_DDIM, help="(profiling only) fewer denoising steps; the metric needs 50")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r03_traffic.json"),
                    help="per-kernel-class HBM bytes per launch from tools/collect_traffic.sh (PMC passes of this command)")
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if args.launcher_selftest:
        return launcher_selftest(world, rank)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)  # RCCL; barrier + max-reduce of the time only

    import mrisr
    from mrisr import _lib as L
    from mrisr import params as P

    log("init weights on device")
    cfg = mrisr.UNetConfig()
    sd = P.random_state_dict(P.unet_param_shapes(cfg), SEED, dev)
    sd.update(P.random_state_dict(P.lora_param_shapes(cfg, 4), SEED + 3, dev))
    unet = mrisr.UNet2DConditionModel(cfg, compute_dtype=args.dtype, lora_rank=4, lora_alpha=4,
                                      lora_fused=not args.lora_merged, flash_attention=True)
    log("load_state_dict + finalize")
    unet.load_state_dict(sd)
    torch.cuda.synchronize()
    log("weights packed")
    sched = mrisr.DDIMScheduler(timestep_spacing="leading", steps_offset=1)
    sched.set_timesteps(args.ddim_steps)
    sampler = mrisr.Sampler(unet, sched, kind="ddim")
    B = args.batch
    lr_lat, ctx, noise, hr = synthetic_batch(B, dev, rank)
    a_T = float(sched.alphas_cumprod[int(sched.timesteps[0])])
    x_T = (lr_lat + (1 - a_T) ** 0.5 * noise).contiguous()  # reference res_srdiff.py:58
    lat = torch.empty_like(x_T)

    def one_batch():
        lat.copy_(x_T)
        sampler.run(lat, ctx, use_graph=not args.no_graph)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_batch()
        torch.cuda.synchronize()
        log(f"warmup batch {i} done")
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        one_batch()
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed:.3f} s for {args.steps} batches")
    from mrisr import dist as mdist
    elapsed = mdist.max_over_ranks(elapsed, dev)  # slowest rank's wall time
    finite = bool(torch.isfinite(lat).all())
    final_slice0 = lat[:1].detach().float().cpu()  # the timed runs' result for slice 0 (the roofline leg below reuses `lat`)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline leg: the same workload, eager launches, a HIP event pair around EVERY kernel launch on the launch
    # stream (libmrisr's own profiler), aggregated per kernel class; two denoising steps.
    lib = L.lib()
    lib.mrisr_prof_reset()
    lib.mrisr_prof_enable(1)
    sampler2 = mrisr.Sampler(unet, sched, kind="ddim")
    sampler2.set_range(0, 2)
    lat.copy_(x_T)
    sampler2.run(lat, ctx, use_graph=False)
    torch.cuda.synchronize()
    lib.mrisr_prof_enable(0)
    classes = prof_report(lib)
    lib.mrisr_prof_reset()
    log("roofline leg done")
    total_ms = sum(v["ms"] for v in classes.values()) or 1.0
    gemm = {k: v for k, v in classes.items() if k.startswith("gemm_")}
    dom_name = max(gemm, key=lambda k: gemm[k]["ms"]) if gemm else None
    roof = None
    if dom_name:
        d = gemm[dom_name]
        per_launch_ms = d["ms"] / d["launches"]
        achieved = d["flops"] / d["launches"] / (per_launch_ms * 1e-3) / 1e12
        all_fl = sum(v["flops"] for v in gemm.values())
        all_ms = sum(v["ms"] for v in gemm.values())
        roof = {"bound": "mfma", "kernel": dom_name, "achieved": achieved, "peak": PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3,
                "unit": "TFLOP/s", "frac": achieved / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3), "traffic": None,
                "launches": d["launches"], "avg_launch_us": per_launch_ms * 1e3,
                "alg_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                "share_of_step_time": d["ms"] / total_ms,
                "all_gemm": {"achieved": all_fl / (all_ms * 1e-3) / 1e12, "share_of_step_time": all_ms / total_ms,
                             "alg_gflop_per_unet_step": all_fl / 2 / 1e9},
                "classes_ms_per_step": {k: round(v["ms"] / 2, 4) for k, v in sorted(classes.items(), key=lambda kv: -kv[1]["ms"])}}
        try:  # HBM bytes per launch of the dominant kernel, from the committed PMC collection of this same command
            tj = json.load(open(args.traffic_json))
            if dom_name in tj:
                roof["traffic"] = tj[dom_name]["hbm_bytes_per_launch"]
                roof["traffic_source"] = (f"{os.path.relpath(args.traffic_json, ROOT)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                          "command (tools/collect_traffic.sh), NOT collected in this run")
                roof["alg_bytes_per_launch"] = d["bytes"] / d["launches"]
        except (OSError, ValueError):
            pass

    total_slices = world * B * args.steps
    value = total_slices / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    step_ms = ms_per_step / args.ddim_steps
    out = {
        "metric": "MRI slices/sec (50-step DDIM, 256^2, bs=32)", "value": value, "unit": "slices/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "configs[1]: 256x256 1-ch synthetic MRI slices -> 4x32x32 latents, SD-1.5-size UNet "
                               "(859.5M params, random init) + rank-4 LoRA, 50-step DDIM, bs=32 per GPU",
                   "weights": "random-init (no checkpoints offline); the fidelity block is numerical agreement with the CPU oracle only, "
                              "it says nothing about image quality",
                   "slices_per_gpu_per_step": B, "ddim_steps": args.ddim_steps, "parallelism": f"slice-sharded x{world} (no collective)",
                   "lora": "merged" if args.lora_merged else "explicit adapters, down-projection + rank-r update inside the projection GEMMs",
                   "tile_table": (os.path.relpath(os.environ["MRISR_TUNE_CACHE"], ROOT) if os.environ.get("MRISR_TUNE_CACHE") else "online autotune"),
                   "hipgraph": not args.no_graph},
        "denoise_step_ms": step_ms,
        "unet_tflops_end_to_end": UNET_GFLOP_PER_SAMPLE * B / step_ms,
        "frac_of_bf16_peak_end_to_end": UNET_GFLOP_PER_SAMPLE * B / step_ms / PEAK_BF16_TFLOPS,
        "finite": finite,
        "roofline": roof,
    }
    if world == 1 and not args.no_cpu_baseline:
        from oracle import unet as ou
        threads = host_threads()
        # ---- the metric's second half ("PSNR vs ref"): slice 0 of this batch through the CPU oracle for all the DDIM steps,
        # against (a) the bf16 result of the timed runs and (b) the f32 device engine on the same slice
        bf16_lat = final_slice0
        f32_lat = None
        if args.dtype == "bf16":
            log("f32 device engine on slice 0 (parity leg)")
            u32 = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4, lora_fused=not args.lora_merged,
                                             flash_attention=True)
            u32.load_state_dict(sd)
            l32 = x_T[:1].clone().contiguous()
            mrisr.Sampler(u32, sched, kind="ddim").run(l32, ctx[:1], use_graph=False)
            torch.cuda.synchronize()
            f32_lat = l32.float().cpu()
            del u32
        log(f"cpu oracle: {args.ddim_steps}-step DDIM of slice 0 on {threads} threads")
        sd_cpu = {k: v.detach().float().cpu() for k, v in sd.items()}
        out["cpu_baseline"], ref_lat = cpu_baseline(sd_cpu, ou.SD15, threads, x_T[:1].float().cpu(), ctx[:1].float().cpu(),
                                                    args.ddim_steps)
        hr01 = (hr[:1].float().cpu() / 2 + 0.5).clamp(0, 1)
        ref_img = stub_decode(ref_lat)

        def score(z):
            img = stub_decode(z)
            return {"rel_l2_latents_vs_oracle": float((z - ref_lat).norm() / ref_lat.norm()),
                    "rel_l2_image_vs_oracle": float((img - ref_img).norm() / ref_img.norm()),
                    "psnr_vs_oracle_db": psnr(img, ref_img), "psnr_vs_hr_db": psnr(img, hr01)}

        fid = {"slice": "rank 0, slice 0, stub-decoded [0,1] 256x256", "oracle_psnr_vs_hr_db": psnr(ref_img, hr01),
               args.dtype: score(bf16_lat)}
        if f32_lat is not None:
            fid["f32"] = score(f32_lat)
        fid["psnr_diff_vs_oracle_db"] = abs(fid[args.dtype]["psnr_vs_hr_db"] - fid["oracle_psnr_vs_hr_db"])
        out["fidelity"] = fid
        out["psnr_vs_oracle_db"] = fid[args.dtype]["psnr_vs_oracle_db"]
        out["rel_l2_f32_vs_oracle"] = (fid.get("f32") or fid[args.dtype])["rel_l2_latents_vs_oracle"]
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
