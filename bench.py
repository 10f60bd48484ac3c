#!/usr/bin/env python
"""bench.py -- denoising steps/sec of the Ctrl-V hot path on MI355X (BASELINE.json metric).

One "step" = one scheduler iteration for one 25-frame clip at 576x1024 (latent 25x4x72x128) with CFG:
ControlNet forward + UNet forward (batch 2) + CFG combine + Euler update
(/root/reference/src/ctrlv/pipelines/pipeline_video_control.py:298-343), on synthetic seeded inputs and random-init
weights of the SVD-XT architecture, inputs resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Multi-GPU: clips shard one-per-rank, weights replicated, NO collective inside the loop (weak scaling); a barrier +
torch.cuda.synchronize() brackets the timed region and the elapsed time is the MAX over ranks.

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel family, HIP-event timed on the
launch stream during the last timed step), `rooflines` (every family, incl. the attention-MFMA and GroupNorm-HBM
fractions north_star asks for) and `cpu_baseline` (the CPU oracle timed on this host on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_MFMA_BF16_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16
PEAK_HBM_GBS = 8000.0               # 8.0 TB/s spec (6.29 TB/s measured float4 copy)
ALG_TFLOP_PER_STEP = {"box2video": 218.52, "svd_unet": 159.90}     # SURVEY.md Appendix B (CFG, 25x72x128)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--workload", choices=["box2video", "svd_unet"], default="box2video",
                    help="box2video = BASELINE configs[2] (SVD + ControlNet, the metric's config); svd_unet = configs[1]")
    ap.add_argument("--hip-graph", type=int, default=int(os.environ.get("CTRLV_HIP_GRAPH", "1")),
                    help="replay the two model forwards from a captured HIP graph (default on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="debug: only time the CPU oracle sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU oracle (0 = min(cores, 32))")
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--cpu-latent", type=int, default=32)
    ap.add_argument("--seed", type=int, default=1234)
    return ap.parse_args()


def build_models(device, workload, frames):
    from ctrlv_amd.models import ControlNetModel, UNetSpatioTemporalConditionModel
    from ctrlv_amd.utils import build_on_device, random_init_
    unet = build_on_device(UNetSpatioTemporalConditionModel, device, num_frames=frames)
    random_init_(unet, seed=0)
    ctrl = None
    if workload == "box2video":
        ctrl = build_on_device(ControlNetModel, device, num_frames=frames)
        random_init_(ctrl, seed=1, zero_conv_std=0.02)
    return unet, ctrl


def make_stepper(unet, ctrl, device, args, clip_index):
    """Synthetic inputs of SURVEY.md 8(d): seeded N(0,1) latents * init_noise_sigma, image latents / control latents /
    CLIP embedding with a zero unconditional half, added ids [6, 127, 0.02], guidance linspace(1, 3, F)."""
    from ctrlv_amd.distributed import clip_generator
    from ctrlv_amd.pipelines.pipeline_utils import DenoiseStepper
    from ctrlv_amd.schedulers import EulerDiscreteScheduler
    g = clip_generator(args.seed, clip_index)
    F, h, w = args.frames, args.height // 8, args.width // 8
    sched = EulerDiscreteScheduler()
    sched.set_timesteps(25, device=device)
    bf = torch.bfloat16
    latents = (torch.randn(1, F, 4, h, w, generator=g) * sched.init_noise_sigma).to(device)
    img = torch.randn(1, 4, h, w, generator=g)
    image_latents = torch.cat([torch.zeros_like(img), img]).unsqueeze(1).repeat(1, F, 1, 1, 1).to(device, bf)
    e = torch.randn(1, 1, 1024, generator=g)
    ehs = torch.cat([torch.zeros_like(e), e]).to(device, bf)
    c = torch.randn(1, F, 4, h, w, generator=g)
    cond = torch.cat([torch.zeros_like(c), c]).to(device, bf) if ctrl is not None else None
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2, device=device, dtype=bf)
    st = DenoiseStepper(unet, ctrl, sched, latents, image_latents, ehs, ids, cond, 1.0, 3.0, 1.0, do_cfg=True,
                        use_hip_graph=bool(args.hip_graph))
    st._init_latents = latents.clone()
    return st


def run_step(st, i):
    k = i % 25
    if k == 0 and i > 0:                       # wrapped around the 25-step schedule: restart the clip
        st.set_latents(st._init_latents, 0)
    st.step(k)


def cpu_baseline(args):
    """The CPU oracle (full SVD widths, fp32, all host cores) on a bounded sample: ONE ControlNet + UNet forward at
    B=1 (no CFG), `cpu_frames` frames, `cpu_latent`^2 latent.  Converted to the metric's unit by algorithmic FLOPs:
    steps/s-equivalent = (sample TFLOP / seconds) / (218.52 TFLOP per full step)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctrlv_ref as R
    cores = args.cpu_threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    F, L = args.cpu_frames, args.cpu_latent
    with torch.no_grad():
        with torch.device("meta"):
            unet = R.UNetSpatioTemporalConditionModel(num_frames=F)
            ctrl = R.ControlNetModel(num_frames=F) if args.workload == "box2video" else None
        src = torch.rand(1 << 24) * 2 - 1
        for m in (unet, ctrl):
            if m is None:
                continue
            m.to_empty(device="cpu")
            for name, p in m.named_parameters():
                n = p.numel()
                flat = p.view(-1)
                for o in range(0, n, src.numel()):
                    k = min(src.numel(), n - o)
                    flat[o:o + k].copy_(src[:k])
                if name.endswith(("norm.weight", "norm1.weight", "norm2.weight", "norm3.weight", "norm_in.weight",
                                  "conv_norm_out.weight")):
                    p.fill_(1.0)
                elif name.endswith("mix_factor"):
                    p.fill_(0.5)
                elif p.dim() > 1:
                    p.mul_(1.0 / p[0].numel() ** 0.5)
                else:
                    p.mul_(0.02)
            m.eval()
        g = torch.Generator().manual_seed(0)
        sample = torch.randn(1, F, 8, L, L, generator=g)
        cond = torch.randn(1, F, 4, L, L, generator=g)
        ehs = torch.randn(1, 1, 1024, generator=g)
        ids = torch.tensor([[6.0, 127.0, 0.02]])
        t = torch.tensor(1.6377)

        def fwd():
            down = mid = None
            if ctrl is not None:
                down, mid = ctrl(sample, t, ehs, ids, control_cond=cond)
            return unet(sample, t, ehs, ids, down, mid)[0]

        fwd()                                   # warm-up (page-in, thread pool)
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            out = fwd()
            times.append(time.perf_counter() - t0)
        assert torch.isfinite(out).all()
    sec = min(times)
    # algorithmic FLOPs of the sample: everything scales with pixels*frames except spatial attention (quadratic in S)
    full = ALG_TFLOP_PER_STEP[args.workload]
    attn_full = 43.41 if args.workload == "box2video" else 31.01
    px = (F * L * L) / (50.0 * 72 * 128)
    tf = (full - attn_full) * px + attn_full * px * (L * L) / (72.0 * 128)
    try:
        model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": round(tf / sec / full, 6), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle (plain PyTorch fp32, {cores} threads, {model}) ControlNet+UNet forward, full SVD widths, "
                      f"B=1 no CFG, {F} frames, {L}x{L} latent = {tf:.3f} TFLOP in {sec:.2f} s "
                      f"({tf / sec * 1e3:.0f} GFLOP/s); scaled to 218.52 TFLOP/step"}


def cpu_baseline_child(args):
    """Runs the CPU-oracle timing in a child process (own thread pool, hard time limit) so that a pathological host
    (e.g. 256 hardware threads oversubscribing small ops) cannot stall the benchmark; retries with fewer threads."""
    import subprocess
    total = os.cpu_count() or 1
    for thr in ([args.cpu_threads] if args.cpu_threads else [min(total, 32), 8]):
        env = dict(os.environ, OMP_NUM_THREADS=str(thr), MKL_NUM_THREADS=str(thr))
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-threads", str(thr),
               "--cpu-frames", str(args.cpu_frames), "--cpu-latent", str(args.cpu_latent), "--workload", args.workload]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
            for ln in reversed(r.stdout.strip().splitlines()):
                if ln.startswith("{"):
                    res = json.loads(ln)
                    res["host_cores"] = total
                    return res
            log(f"cpu baseline child ({thr} threads) produced no result: {r.stderr[-300:]}")
        except subprocess.TimeoutExpired:
            log(f"cpu baseline child with {thr} threads exceeded 240 s")
    return {"value": None, "unit": "steps/s", "cores": 0, "kind": "port", "sample": "CPU oracle timing failed on this host"}


def log(*a):
    print(f"[bench {time.strftime('%H:%M:%S')}]", *a, file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)))
        return
    from ctrlv_amd import distributed as D
    from ctrlv_amd import profiler
    rank, world, local = D.init("nccl" if int(os.environ.get("WORLD_SIZE", 1)) > 1 else None)
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    unet, ctrl = build_models(device, args.workload, args.frames)
    st = make_stepper(unet, ctrl, device, args, clip_index=rank)       # one clip per rank (weak scaling)
    log(f"rank {rank}/{world}: models built on {device}, hip_graph={args.hip_graph}")

    i = 0
    for _ in range(max(args.warmup, 2 if args.hip_graph else 1)):      # graph mode: 1 eager + 1 capture step
        run_step(st, i); i += 1
    torch.cuda.synchronize()
    log(f"warm-up done; workspace peak {unet._ws.peak / 2**30:.1f} GiB (UNet)"
        + (f" + {ctrl._ws.peak / 2**30:.1f} GiB (ControlNet)" if ctrl is not None else "")
        + f"; torch allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB")
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    timer = None
    for k in range(args.steps):
        if k == args.steps - 1 and rank == 0 and not args.hip_graph:
            timer = profiler.KernelTimer()
            with timer:
                run_step(st, i)
        else:
            run_step(st, i)
        i += 1
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    finite = bool(torch.isfinite(st.latents).all())
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s")

    if rank == 0 and args.hip_graph:
        # per-kernel HIP-event timing needs eager launches: one extra identical step outside the timed region
        st.use_hip_graph = False
        timer = profiler.KernelTimer()
        with timer:
            run_step(st, i)
        torch.cuda.synchronize()

    if rank != 0:
        return
    fams = timer.summary() if timer is not None else {}
    # HBM traffic per launch from the separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, gfx950 corrections
    # applied by tools/pmc_summary.py) committed under profiles/; null when no summary is present
    pmc = {}
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic_summary.json")))
        if cands and args.workload == "box2video" and (args.height, args.width, args.frames) == (576, 1024, 25):
            raw = json.load(open(cands[-1]))
            alias = {"attention_spatial": ["attn_spatial_kernel"], "attention_temporal": ["attn_temporal_kernel"],
                     "groupnorm": ["gn_stats_kernel", "gn_apply_kernel"], "layernorm": ["ln_kernel"],
                     "residual_add": ["axpby_kernel"], "gemm_linear": ["gemm_linear"],
                     "gemm_conv3x3": ["gemm_conv3x3"], "gemm_conv_temporal": ["gemm_conv_temporal"]}
            for fam, keys in alias.items():
                if all(k in raw for k in keys):
                    pmc[fam] = sum(raw[k]["traffic_bytes_per_launch"] for k in keys)
            pmc["_source"] = os.path.relpath(cands[-1], ROOT)
    except Exception as ex:       # noqa: BLE001
        log(f"no PMC traffic summary: {ex}")
    rooflines = {}
    for fam, d in fams.items():
        sec = d["ms"] * 1e-3
        if d["flops"] > 0 and fam.startswith(("gemm", "attention_spatial")):
            ach = d["flops"] / sec / 1e12
            rooflines[fam] = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_MFMA_BF16_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_BF16_TFLOPS, 4),
                              "traffic": pmc.get(fam),
                              "ms_per_step": round(d["ms"], 3), "launches": d["calls"],
                              "avg_launch_ms": round(d["ms"] / d["calls"], 4)}
        else:
            ach = d["bytes"] / sec / 1e9
            rooflines[fam] = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": pmc.get(fam),
                              "ms_per_step": round(d["ms"], 3), "launches": d["calls"],
                              "avg_launch_ms": round(d["ms"] / d["calls"], 4)}
    dominant = max(rooflines, key=lambda f: rooflines[f]["ms_per_step"]) if rooflines else None
    roofline = dict(rooflines[dominant], kernel=dominant) if dominant else None
    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps * world / elapsed
    alg = ALG_TFLOP_PER_STEP[args.workload]
    if (args.height, args.width, args.frames) != (576, 1024, 25):      # table above is for the BASELINE shape only
        alg = round(sum(d["flops"] for d in fams.values()) / 1e12, 2)
    line = {
        "metric": "denoising steps/sec, SVD+ControlNet 25f 576x1024" if args.workload == "box2video"
        else "denoising steps/sec, SVD UNet-only 25f 576x1024",
        "value": round(value, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {'ControlNet + ' if ctrl is not None else ''}UNet forward, CFG batch 2, "
                               f"{args.frames} frames, latent {args.height // 8}x{args.width // 8}, 25-step Karras Euler "
                               "schedule, random-init SVD-XT weights", "clips_per_gpu": 1, "parallelism": f"clip-shard x{world}",
                   "hip_graph": bool(args.hip_graph)},
        "step_mfma_frac": round(alg / (ms_per_step * 1e-3) / PEAK_MFMA_BF16_TFLOPS, 4),
        "algorithmic_tflop_per_step": alg,
        "executed_tflop_per_step": round(sum(d["flops"] for d in fams.values()) / 1e12, 2),
        "kernel_ms_per_step": round(sum(d["ms"] for d in fams.values()), 2),
        "finite": finite,
        "roofline": roofline, "rooflines": rooflines,
        "traffic_source": pmc.get("_source"),
    }
    if world == 1 and not args.no_cpu_baseline:
        del st, unet, ctrl
        torch.cuda.empty_cache()
        log("timing the CPU oracle baseline (bounded sample) ...")
        line["cpu_baseline"] = cpu_baseline_child(args)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
