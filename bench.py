#!/usr/bin/env python
"""bench.py -- denoising steps/sec of the Ctrl-V hot path on MI355X (BASELINE.json metric).

One "step" = one scheduler iteration for one 25-frame clip at 576x1024 (latent 25x4x72x128) with CFG:
ControlNet forward + UNet forward (batch 2) + CFG combine + Euler update
(/root/reference/src/ctrlv/pipelines/pipeline_video_control.py:298-343), on synthetic seeded inputs and random-init
weights of the SVD-XT architecture, inputs resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE
in the environment), or started plainly -- then this process, WITHOUT touching the GPU, starts the N rank processes
itself and relays rank 0's JSON line.  Clips shard one-per-rank, weights replicated, NO collective inside the loop (weak
scaling); a barrier + torch.cuda.synchronize() brackets the timed region and the elapsed time is the MAX over ranks.
`n_gpus` in the output is the world size the process group reports.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline / rooflines   per kernel family (HIP events on the launch stream during one instrumented eager step), incl.
                         the attention-MFMA and GroupNorm-HBM fractions north_star asks for; `traffic` = PMC HBM bytes
                         per launch from the committed rocprofv3 --pmc summary, or null when that summary was not
                         collected from the library that is running (build-id stamp) or the shape differs;
  cpu_baseline           the CPU oracle timed on this host's cores in the same run: by default ONE COMPLETE no-CFG step
                         (B = 1, all 25 frames at the full latent resolution, full SVD widths; x2 for the CFG pair), the
                         thread count swept up to the physical core count; `--cpu-frames N` bounds it to N frames
                         (converted by the frame count: per-frame work is identical; see cpu_baseline());
  parity                 the HIP models' output on that same sample (same weights, same inputs) against the oracle's.
The CPU baseline times one complete no-CFG step (25 frames, ~109 TFLOP, ~2 minutes); `--cpu-frames 2` bounds it to a ~10 s sample.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU oracle (0 = calibrate)")
    ap.add_argument("--cpu-frames", type=int, default=0,
                    help="frames of the CPU baseline's sample at the full latent size; 0 (default) = ALL frames: the "
                         "complete no-CFG step (~109 TFLOP, ~2 minutes on the box's host cores); e.g. 2 for a ~10 s sample")
    ap.add_argument("--cpu-full-step", action="store_true", help="(default since round 3; kept for old command lines)")
    ap.add_argument("--cpu-timeout", type=int, default=0, help="seconds (0 = 240 for a sample, 1500 for the full step)")
    ap.add_argument("--cpu-baseline-child", default="", help=argparse.SUPPRESS)      # internal: exchange directory
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU work: ranks rendezvous over gloo and run the barrier / MAX-reduce bracket only "
                         "(checks the N > 1 launch path on a machine without GPUs; prints value = null)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16",
                    help="element type of activations / packed weights: bf16 (BASELINE.json's dtype, libctrlv_hip.so) or "
                         "fp16 (the reference's autocast dtype, libctrlv_hip_f16.so: model-level rel-L2 1.3e-3 measured instead of 1.0e-2, "
                         "gate 2e-3; below north_star's 1e-3 with --trunk fp16x2)")
    ap.add_argument("--trunk", choices=["same", "fp16x2"], default="same",
                    help="storage of the residual trunk (with --dtype fp16): same = one fp16 element per value; fp16x2 = split "
                         "fp16 hi + one-byte e5m2 lo planes (~15 significant bits in 3 bytes) -- north_star's 1e-3 model-level tolerance")
    ap.add_argument("--no-profile-step", action="store_true",
                    help="skip the extra instrumented eager step (no `roofline` in the output): for runs under rocprofv3 --pmc")
    ap.add_argument("--no-fp16-leg", action="store_true",
                    help="bf16 runs: skip the extra leg that runs the SAME weights / sample through the fp16 build "
                         "(its parity and step time are reported as `fp16_build` beside the bf16 headline)")
    return ap.parse_args()


def torch_dtype(name):
    import torch
    return torch.float16 if name == "fp16" else torch.bfloat16


def log(*a):
    print(f"[bench {time.strftime('%H:%M:%S')}]", *a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------- multi-GPU launch
def launch_ranks(args):
    """`--gpus N` without a torchrun environment: N rank processes from this (GPU-free) process
    (ctrlv_amd.distributed.launch_local_ranks; importing it initialises no device)."""
    sys.path.insert(0, ROOT)
    from ctrlv_amd.distributed import launch_local_ranks
    launch_local_ranks(__file__, sys.argv[1:], args.gpus, log)


# ---------------------------------------------------------------------------------------------------- GPU side
def build_models(device, workload, frames, dtype):
    """Random-init SVD-XT weights, created in bf16 (the benchmark's weights whatever the element type: an fp16 run holds the
    same values -- bf16 -> fp16 is exact for every weight above fp16's subnormal range)."""
    import torch
    from ctrlv_amd.models import ControlNetModel, UNetSpatioTemporalConditionModel
    from ctrlv_amd.utils import build_on_device, random_init_
    unet = build_on_device(UNetSpatioTemporalConditionModel, device, num_frames=frames)
    random_init_(unet, seed=0)
    ctrl = None
    if workload == "box2video":
        ctrl = build_on_device(ControlNetModel, device, num_frames=frames)
        random_init_(ctrl, seed=1, zero_conv_std=0.02)
    if dtype != torch.bfloat16:
        unet.to(dtype)
        if ctrl is not None:
            ctrl.to(dtype)
    return unet, ctrl


def make_stepper(unet, ctrl, device, args, clip_index):
    """Synthetic inputs of SURVEY.md 8(d): seeded N(0,1) latents * init_noise_sigma, image latents / control latents /
    CLIP embedding with a zero unconditional half, added ids [6, 127, 0.02], guidance linspace(1, 3, F)."""
    import torch
    from ctrlv_amd.distributed import clip_generator
    from ctrlv_amd.pipelines.pipeline_utils import DenoiseStepper
    from ctrlv_amd.schedulers import EulerDiscreteScheduler
    g = clip_generator(args.seed, clip_index)
    F, h, w = args.frames, args.height // 8, args.width // 8
    sched = EulerDiscreteScheduler()
    sched.set_timesteps(25, device=device)
    bf = unet.dtype
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
        st.set_latents(st._init_latents, 0)    # (set_latents copies: the initial latents are never overwritten)
    st.step(k)


# ---------------------------------------------------------------------------------------------------- CPU baseline
def cpu_sample_inputs(F, h, w, seed=0):
    import torch
    g = torch.Generator().manual_seed(seed)
    # bf16-representable, hence fp16-representable too (8 bits of mantissa; N(0,1) values below fp16's normal range differ
    # by < 3e-8): both sides, both element builds, see identical inputs
    q = lambda x: x.to(torch.bfloat16).float()   # noqa: E731
    return dict(sample=q(torch.randn(1, F, 8, h, w, generator=g)), cond=q(torch.randn(1, F, 4, h, w, generator=g)),
                ehs=q(torch.randn(1, 1, 1024, generator=g)), ids=torch.tensor([[6.0, 127.0, 0.02]]),
                t=torch.tensor(1.6377))


def cpu_baseline_child(xdir):
    """Child process: time the CPU oracle (ctrlv_ref, plain PyTorch fp32) on the sample described by <xdir>/job.json with
    the weights in <xdir>/{unet,controlnet}.safetensors (the GPU run's own bf16 weights, up-cast), write the outputs
    and the timing back.  The oracle is the checker / reported baseline here, never the measured product path."""
    import torch
    from safetensors.torch import load_file
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctrlv_ref as R
    job = json.load(open(os.path.join(xdir, "job.json")))
    F, h, w = job["frames"], job["h"], job["w"]
    threads = job["threads"]
    if not threads:
        # calibrate on the workload's own shape (a 1-frame forward at the FULL latent: convs, GEMMs and the 9216-token
        # attention at their real per-frame size), sweeping the thread count up to the PHYSICAL core count; the fastest wins
        total = os.cpu_count() or 1
        try:
            sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read()
            smt = max(1, len([x for x in sib.replace("-", ",").split(",") if x.strip()]))
        except Exception:       # noqa: BLE001
            smt = 2
        phys = max(1, total // smt)
        job["physical_cores"] = phys
        job["calibrate"] = sorted({min(phys, n) for n in (16, 32, 64, 128, phys)})
        threads = 0
    if threads:
        torch.set_num_threads(threads)
    with torch.no_grad():
        with torch.device("meta"):
            unet = R.UNetSpatioTemporalConditionModel(num_frames=F)
            ctrl = R.ControlNetModel(num_frames=F) if job["workload"] == "box2video" else None
        for m, name in ((unet, "unet"), (ctrl, "controlnet")):
            if m is None:
                continue
            sd = {k: v.float() for k, v in load_file(os.path.join(xdir, name + ".safetensors")).items()}
            m.load_state_dict(sd, assign=True)
            m.eval()
        inp = cpu_sample_inputs(F, h, w)

        def fwd(x):
            down = mid = None
            if ctrl is not None:
                down, mid = ctrl(x["sample"], x["t"], x["ehs"], x["ids"], control_cond=x["cond"])
            return unet(x["sample"], x["t"], x["ehs"], x["ids"], down, mid)[0]

        fwd(cpu_sample_inputs(F, 8, 8))          # thread-pool / allocator warm-up on a tiny latent
        sweep = {}
        if not threads:
            cal = cpu_sample_inputs(1, h, w)         # ONE frame at the FULL latent: the per-frame work of the timed sample
            for thr in job["calibrate"]:
                torch.set_num_threads(thr)
                t0 = time.perf_counter()
                fwd(cal)
                sweep[thr] = round(time.perf_counter() - t0, 3)
            threads = min(sweep, key=sweep.get)
            torch.set_num_threads(threads)
        t0 = time.perf_counter()
        out = fwd(inp)
        sec = time.perf_counter() - t0
    torch.save(out, os.path.join(xdir, "out.pt"))
    try:
        model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:       # noqa: BLE001
        model = "unknown"
    json.dump({"seconds": sec, "threads": threads, "cpu": model, "host_threads": os.cpu_count(),
               "physical_cores": job.get("physical_cores"), "thread_sweep_seconds": sweep},
              open(os.path.join(xdir, "result.json"), "w"))


def cpu_baseline(args, unet, ctrl, device):
    """cpu_baseline + parity legs.

    Sample: B = 1 (no CFG), F = --cpu-frames frames at the FULL latent resolution and full SVD widths, i.e. the benchmark
    workload with fewer frames -- spatial attention (quadratic in the 9216 tokens), convs, GEMMs and norms all run at
    their real per-frame size.  Conversion to the metric's unit: one benchmark step is 2 x 25 = 50 frame-images, the
    sample is F of them, every kernel's work is per frame (the only cross-frame operators, the 3-tap temporal conv and
    the F x F temporal attention, are <= 7 % of the FLOPs and linear / negligible in F), so
        steps/s = 1 / (seconds * 50 / F).
    With --cpu-full-step the sample is the complete 25-frame no-CFG forward pair and the factor is exactly 2 (CFG).
    The oracle runs in a child process (own thread pool, hard time limit) on the SAME weights as the GPU models (the
    bf16 parameters are handed over through a tmpfs directory) and the SAME inputs; the HIP output on that sample is
    compared with the oracle's -> `parity`."""
    import torch
    from safetensors.torch import save_file
    full = args.cpu_full_step or args.cpu_frames <= 0 or args.cpu_frames >= args.frames
    F = args.frames if full else args.cpu_frames
    h, w = args.height // 8, args.width // 8
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    xdir = tempfile.mkdtemp(prefix="ctrlv_bench_", dir=base)
    try:
        for m, name in ((unet, "unet"), (ctrl, "controlnet")):
            if m is not None:
                save_file({k: v.detach().cpu().contiguous() for k, v in m.state_dict().items()},
                          os.path.join(xdir, name + ".safetensors"))
        json.dump({"frames": F, "h": h, "w": w, "threads": args.cpu_threads, "workload": args.workload},
                  open(os.path.join(xdir, "job.json"), "w"))
        timeout = args.cpu_timeout or (1500 if full else 240)
        env = dict(os.environ)
        env.pop("OMP_NUM_THREADS", None)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", xdir], env=env,
                               capture_output=True, text=True, timeout=timeout)
        except subprocess.TimeoutExpired:
            return ({"value": None, "unit": "steps/s", "cores": 0, "kind": "port",
                     "sample": f"CPU oracle sample ({F} frames at {h}x{w}) exceeded {timeout} s on this host"}, None, None)
        if r.returncode != 0 or not os.path.exists(os.path.join(xdir, "result.json")):
            return ({"value": None, "unit": "steps/s", "cores": 0, "kind": "port",
                     "sample": "CPU oracle child failed: " + r.stderr[-300:]}, None, None)
        res = json.load(open(os.path.join(xdir, "result.json")))
        ref = torch.load(os.path.join(xdir, "out.pt"))
    finally:
        shutil.rmtree(xdir, ignore_errors=True)
    sec = res["seconds"]
    frames_per_step = 2 * args.frames
    value = 1.0 / (sec * frames_per_step / F)
    cpu = {"value": round(value, 6), "unit": "steps/s", "cores": res["threads"], "kind": "port",
           "host_threads": res["host_threads"], "physical_cores": res.get("physical_cores"),
           "thread_sweep_seconds": res.get("thread_sweep_seconds"), "sample_seconds": round(sec, 2),
           "scaled_by": f"{frames_per_step}/{F} frame-images per step / per sample",
           "sample": ("the complete no-CFG step: " if full else "") +
                     f"oracle (plain PyTorch fp32, {res['threads']} threads of {res['cpu']}) "
                     f"{'ControlNet + ' if ctrl is not None else ''}UNet forward, full SVD widths, B=1 no CFG, {F} of the "
                     f"step's {frames_per_step} frame-images at the full {h}x{w} latent, measured {sec:.1f} s"
                     + (" (x2 for the CFG pair)" if full else "")}
    # ---- parity of the HIP path on the same sample
    ref = ref.float()
    parity = hip_parity(unet, ctrl, ref, F, h, w, device)
    return cpu, parity, ref


# Bounds this bench asserts per storage mode (measured on the complete 25-frame step: bf16 1.03e-2, fp16 1.28e-3, fp16 with
# the split trunk < 1e-3) and the number north_star itself states.  `parity.ok` is against `tolerance`;
# `parity.meets_north_star` against NORTH_STAR_TOL.
PARITY_TOL = {"bf16": 1.5e-2, "fp16": 2e-3, "fp16+fp16x2": 1e-3}
NORTH_STAR_TOL = 1e-3


def hip_parity(unet, ctrl, ref, F, h, w, device):
    """The HIP models' output on the CPU-baseline sample against the oracle's (`ref`), in the models' own dtype."""
    import torch
    from tests.parity_utils import max_err, rel_l2
    inp = cpu_sample_inputs(F, h, w)
    el = "fp16" if unet.dtype == torch.float16 else "bf16"
    if getattr(unet, "trunk_dtype", "same") != "same":
        el += "+" + unet.trunk_dtype
    dv = lambda x: x.to(device, unet.dtype)   # noqa: E731
    with torch.no_grad():
        down = mid = None
        if ctrl is not None:
            down, mid = ctrl(dv(inp["sample"]), inp["t"].to(device), dv(inp["ehs"]), inp["ids"].to(device),
                             control_cond=dv(inp["cond"]), return_dict=False)
        got = unet(dv(inp["sample"]), inp["t"].to(device), dv(inp["ehs"]), inp["ids"].to(device), down, mid,
                   return_dict=False)[0].float().cpu()
    rms = ref.pow(2).mean().sqrt().item()
    # one bound for both measures (tests/parity_utils.parity_err): rel-L2 < tol AND every element within
    # tol * (6 rms(ref) + 2 |ref|)
    parity = {"rel_l2": round(rel_l2(got, ref), 6), "max_elem": round(max_err(got, ref), 6),
              "max_abs": round((got - ref).abs().max().item(), 6), "ref_rms": round(rms, 6),
              "tolerance": PARITY_TOL[el], "north_star_tolerance": NORTH_STAR_TOL,
              "what": f"HIP UNet output ({el} storage) vs the fp32 CPU oracle, same weights / inputs, {F} frames at {h}x{w}; "
                      "ok = rel-L2 < tolerance and |a - ref| < tolerance * (6 rms(ref) + 2 |ref|) for every element with "
                      "this storage mode's own bound; meets_north_star = rel-L2 < north_star_tolerance (BASELINE.json: "
                      "'within 1e-3 relative')"}
    parity["ok"] = bool(parity["rel_l2"] < parity["tolerance"] and parity["max_elem"] < parity["tolerance"])
    parity["meets_north_star"] = bool(parity["rel_l2"] < NORTH_STAR_TOL)
    return parity


def fp16_leg(args, unet, ctrl, ref, device, trunk="same"):
    """bf16 runs only: the SAME weights and the SAME sample through the fp16 element build (libctrlv_hip_f16.so) -- its
    parity against the oracle output `ref` (north_star's 1e-3 needs fp16 storage: DESIGN.md 4) and its step time,
    reported beside the bf16 headline; trunk = "fp16x2": the same with the residual trunk split into hi + lo planes.  Not
    part of `value`."""
    import torch
    from ctrlv_amd.utils import build_on_device

    def clone16(m):
        if m is None:
            return None
        cfg = {k: v for k, v in dict(m.config).items() if not k.startswith("_")}
        m2 = build_on_device(type(m), device, torch.bfloat16, **cfg)
        m2.load_state_dict(m.state_dict())
        return m2.to(torch.float16)
    u16, c16 = clone16(unet), clone16(ctrl)
    for m in (u16, c16):
        if m is not None:
            m.trunk_dtype = trunk
    F = args.frames if (args.cpu_full_step or args.cpu_frames <= 0 or args.cpu_frames >= args.frames) else args.cpu_frames
    out = {"parity": hip_parity(u16, c16, ref, F, args.height // 8, args.width // 8, device)}
    st = make_stepper(u16, c16, device, args, clip_index=0)
    i = 0
    for _ in range(2):
        run_step(st, i); i += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(st, i); i += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.update(value=round(args.steps / dt, 4), unit="steps/s", ms_per_step=round(dt / args.steps * 1e3, 2), steps=args.steps,
               finite=bool(torch.isfinite(st.latents).all()),
               what="the same weights, workload and HIP-graph step with fp16 elements"
                    + (" and the split (fp16x2) residual trunk" if trunk != "same" else "")
                    + f" (bench.py --dtype fp16{' --trunk fp16x2' if trunk != 'same' else ''} is the full line)")
    del st, u16, c16
    torch.cuda.empty_cache()
    return out


def launcher_selftest(args, D):
    """The rank bracket of main() with the GPU work removed (gloo): rendezvous, barrier, K no-op steps, MAX-reduce."""
    import torch.distributed as dist
    rank, world, _ = D.init("gloo" if int(os.environ.get("WORLD_SIZE", 1)) > 1 else None)
    world = D.world_size()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} ranks")
    clip = D.shard_clips(world, rank, world)[0]
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    seen = [None] * world
    if dist.is_initialized():
        dist.all_gather_object(seen, clip)
    else:
        seen = [clip]
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (no GPU work)", "value": None, "unit": "steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
                          "launcher_selftest": True, "ranks_seen": sorted(seen)}))
    if dist.is_initialized():
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    if args.cpu_baseline_child:
        cpu_baseline_child(args.cpu_baseline_child)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
        return
    import torch
    from ctrlv_amd import distributed as D
    if args.launcher_selftest:
        launcher_selftest(args, D)
        return
    from ctrlv_amd import _lib
    from ctrlv_amd import profiler
    rank, world, local = D.init("nccl" if int(os.environ.get("WORLD_SIZE", 1)) > 1 else None)
    world = D.world_size()                       # what the process group reports, not what the command line asked for
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} ranks")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    unet, ctrl = build_models(device, args.workload, args.frames, torch_dtype(args.dtype))
    if args.trunk != "same":
        if args.dtype != "fp16":
            raise SystemExit("bench.py: --trunk fp16x2 needs --dtype fp16 (the split planes are fp16 elements)")
        for m in (unet, ctrl):
            if m is not None:
                m.trunk_dtype = args.trunk
    clip = D.shard_clips(world, rank, world)[0]                        # one clip per rank (weak scaling)
    st = make_stepper(unet, ctrl, device, args, clip_index=clip)
    lib = _lib.load(unet.el_dtype)
    log(f"rank {rank}/{world}: models built on {device} ({args.dtype} elements), hip_graph={args.hip_graph}, "
        f"lib build {_lib.build_id(lib)}")

    i = 0
    for _ in range(max(args.warmup, 2 if args.hip_graph else 1)):      # graph mode: 1 eager + 1 capture step
        run_step(st, i); i += 1
    torch.cuda.synchronize()
    def ws_gib(m):
        if m._plan is not None:
            return sum(w.numel() for w in m._plan._ws.values()) / 2**30
        return m._ws.peak / 2**30 if m._ws is not None else 0.0
    log(f"warm-up done ({unet.executor} executor); workspace {ws_gib(unet):.1f} GiB (UNet)"
        + (f" + {ws_gib(ctrl):.1f} GiB (ControlNet)" if ctrl is not None else "")
        + f"; torch allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB")
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    timer = None
    for k in range(args.steps):
        run_step(st, i)
        i += 1
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    finite = bool(torch.isfinite(st.latents).all())
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s")

    if rank == 0 and not args.no_profile_step:
        # per-kernel HIP-event timing needs eager launches: one extra identical step outside the timed region, through the
        # SAME executor as the timed steps -- the C++ plan brackets its own launches (ctrlv_plan_profile), so every dispatch
        # decision of the plan (tile choice, fused feed-forward, column groups) is what the roofline describes
        st.use_hip_graph = False
        timer = profiler.PlanTimer(unet, ctrl) if unet.executor == "plan" else profiler.KernelTimer()
        with timer:
            run_step(st, i)
        torch.cuda.synchronize()

    if rank != 0:
        return
    fams = timer.summary() if timer is not None else {}
    # HBM traffic per launch from the separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, gfx950 corrections
    # applied by tools/pmc_summary.py) committed under profiles/.  Only valid for the library it was collected from:
    # the summary carries that library's build id (hash of csrc/ + include/) and is ignored when it differs.
    pmc, pmc_note, pmc_launches = {}, None, {}
    import glob
    default_shape = args.workload == "box2video" and (args.height, args.width, args.frames) == (576, 1024, 25)
    try:
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic_summary.json")))
        match = [c for c in cands if json.load(open(c)).get("_build_id") == _lib.build_id(lib)]
        if match:
            cands = match
        if not cands:
            pmc_note = "no PMC summary under profiles/"
        elif not default_shape:
            pmc_note = "PMC summary is for the default box2video 25x576x1024 shape only"
        else:
            raw = json.load(open(cands[-1]))
            if raw.get("_build_id") != _lib.build_id(lib) or args.dtype != "bf16":
                pmc_note = (f"{os.path.relpath(cands[-1], ROOT)} was collected from library build "
                            f"{raw.get('_build_id', 'unstamped')} (bf16), running {_lib.build_id(lib)} ({args.dtype}): traffic = null")
            else:
                alias = {"attention_spatial": ["attn_spatial_kernel"], "attention_temporal": ["attn_temporal_kernel"],
                         "groupnorm": ["gn_stats_kernel", "gn_finalize_kernel", "gn_apply_kernel"], "layernorm": ["ln_kernel"],
                         "residual_add": ["axpby_kernel"], "gemm_linear": ["gemm_linear"],
                         "gemm_conv3x3": ["gemm_conv3x3"], "gemm_conv_temporal": ["gemm_conv_temporal"],
                         "gemm_temporal_block": ["temporal_fused_kernel"]}
                for fam, keys in alias.items():
                    keys = [k for k in keys if k in raw]
                    if keys:
                        # `traffic` x `traffic_launches` = the family's PMC bytes per step.  traffic_launches = launches per
                        # step of the kernels the figure covers (the ping-pong + fused GEMM kernels of a family, not its tiny
                        # 2-stage launches); a family of several kernels with DIFFERENT launch counts (GroupNorm: apply x 152,
                        # statistics x 68, finalize x 152) is summed per step first and divided by the largest count
                        if all("launches_per_step" in raw[k] for k in keys):
                            pmc_launches[fam] = max(raw[k]["launches_per_step"] for k in keys)
                            per_step = sum(raw[k]["traffic_bytes_per_launch"] * raw[k]["launches_per_step"] for k in keys)
                            pmc[fam] = per_step / pmc_launches[fam]
                        else:
                            pmc[fam] = sum(raw[k]["traffic_bytes_per_launch"] for k in keys)
                pmc["_source"] = os.path.relpath(cands[-1], ROOT)
    except Exception as ex:       # noqa: BLE001
        pmc_note = f"PMC summary unreadable: {ex}"
    # Matrix-pipe utilisation and clock per MFMA family from the committed counter pass (tools/pmc_mfma_pass.sh:
    # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), clock = GRBM_GUI_ACTIVE / 8 / dispatch time; plus the
    # in-kernel clock of the probe kernels of a diagnostic build): same build-id rule as the HBM traffic above.
    mfma, mfma_note = {}, None
    try:
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma_busy_summary.json")))
        # (several may be committed -- e.g. a build before and after a rewrite: take the one collected from this library)
        match = [c for c in cands if json.load(open(c)).get("_build_id") == _lib.build_id(lib)]
        if match:
            cands = match
        if not cands:
            mfma_note = "no MFMA-busy summary under profiles/"
        else:
            raw = json.load(open(cands[-1]))
            if raw.get("_build_id") != _lib.build_id(lib) or args.dtype != "bf16" or not default_shape:
                mfma_note = (f"{os.path.relpath(cands[-1], ROOT)} was collected from library build "
                             f"{raw.get('_build_id', 'unstamped')} (bf16, default shape), running {_lib.build_id(lib)} "
                             f"({args.dtype}): mfma_busy / clock_ghz = null")
            else:
                for fam in ("gemm_linear", "gemm_conv3x3", "gemm_conv_temporal", "attention_spatial", "gemm_temporal_block"):
                    parts = {k.split(".", 1)[1]: v for k, v in raw.items() if k.startswith(fam + ".") and isinstance(v, dict)}
                    if parts:
                        busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in parts.values())
                        act = sum(v["GRBM_GUI_ACTIVE"] for v in parts.values())
                        ms = sum(v["ms"] for v in parts.values())
                        mfma[fam] = {"mfma_busy": round(busy / (act / 8 * 1024), 4), "clock_ghz": round(act / 8 / (ms * 1e-3) / 1e9, 3),
                                     "mfma_busy_by_kernel": {k: v["mfma_busy"] for k, v in parts.items()}}
                mfma["_source"] = os.path.relpath(cands[-1], ROOT)
                if isinstance(raw.get("in_kernel_clock"), (dict, list)):
                    mfma["_in_kernel_clock"] = raw["in_kernel_clock"]
    except Exception as ex:       # noqa: BLE001
        mfma_note = f"MFMA-busy summary unreadable: {ex}"
    rooflines = {}
    for fam, d in fams.items():
        sec = d["ms"] * 1e-3
        if d["flops"] > 0 and fam.startswith(("gemm", "attention_spatial")):
            ach = d["flops"] / sec / 1e12
            rooflines[fam] = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_MFMA_BF16_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_BF16_TFLOPS, 4),
                              "traffic": pmc.get(fam), "traffic_launches": pmc_launches.get(fam),
                              "ms_per_step": round(d["ms"], 3), "launches": d["calls"],
                              "avg_launch_ms": round(d["ms"] / d["calls"], 4),
                              "mfma_busy": mfma.get(fam, {}).get("mfma_busy"), "clock_ghz": mfma.get(fam, {}).get("clock_ghz"),
                              "mfma_busy_by_kernel": mfma.get(fam, {}).get("mfma_busy_by_kernel")}
        else:
            ach = d["bytes"] / sec / 1e9
            rooflines[fam] = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": pmc.get(fam),
                              "traffic_launches": pmc_launches.get(fam),
                              "ms_per_step": round(d["ms"], 3), "launches": d["calls"],
                              "avg_launch_ms": round(d["ms"] / d["calls"], 4)}
    dominant = max(rooflines, key=lambda f: rooflines[f]["ms_per_step"]) if rooflines else None
    roofline = dict(rooflines[dominant], kernel=dominant) if dominant else None
    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps * world / elapsed
    executed = round(sum(d["flops"] for d in fams.values()) / 1e12, 2)
    default_shape = (args.height, args.width, args.frames) == (576, 1024, 25)
    alg = ALG_TFLOP_PER_STEP[args.workload] if default_shape else None      # Appendix B table: BASELINE shape only
    shape = f"{args.frames}f {args.height}x{args.width}"
    line = {
        "metric": (f"denoising steps/sec, SVD+ControlNet {shape}" if args.workload == "box2video"
                   else f"denoising steps/sec, SVD UNet-only {shape}"),
        "value": round(value, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "trunk": args.trunk, "data": "synthetic",
        "config": {"workload": f"{args.workload}: {'ControlNet + ' if ctrl is not None else ''}UNet forward, CFG batch 2, "
                               f"{args.frames} frames, latent {args.height // 8}x{args.width // 8}, 25-step Karras Euler "
                               "schedule, random-init SVD-XT weights", "clips_per_gpu": 1, "parallelism": f"clip-shard x{world}",
                   "hip_graph": bool(args.hip_graph)},
        # whole-step MFMA fraction on the FLOPs the kernels execute (the dead 1-key cross-attention work of the
        # reference, 8.4 TFLOP at the default shape, is not executed and not counted); the reference-algorithm figure
        # is reported next to it
        "step_mfma_frac": round(executed / (ms_per_step * 1e-3) / PEAK_MFMA_BF16_TFLOPS, 4),
        "executed_tflop_per_step": executed,
        "reference_algorithm_tflop_per_step": alg,
        "step_mfma_frac_reference_algorithm": (round(alg / (ms_per_step * 1e-3) / PEAK_MFMA_BF16_TFLOPS, 4)
                                               if alg else None),
        "kernel_ms_per_step": round(sum(d["ms"] for d in fams.values()), 2),
        "finite": finite,
        "lib_build_id": _lib.build_id(lib),
        "roofline": roofline, "rooflines": rooflines,
        "traffic_source": pmc.get("_source"), "traffic_note": pmc_note,
        "mfma_busy_source": mfma.get("_source"), "mfma_busy_note": mfma_note, "in_kernel_clock": mfma.get("_in_kernel_clock"),
    }
    if world == 1 and not args.no_cpu_baseline:
        del st
        torch.cuda.empty_cache()
        log("timing the CPU oracle baseline (same weights; by default the complete no-CFG step, ~2-4 minutes) and checking the HIP output against it ...")
        line["cpu_baseline"], line["parity"], ref = cpu_baseline(args, unet, ctrl, device)
        if args.dtype == "bf16" and ref is not None and not args.no_fp16_leg:
            log("fp16 leg: the same weights / sample through libctrlv_hip_f16.so ...")
            line["fp16_build"] = fp16_leg(args, unet, ctrl, ref, device)
            log("fp16 leg with the split (fp16x2) residual trunk ...")
            line["fp16_split_trunk_build"] = fp16_leg(args, unet, ctrl, ref, device, trunk="fp16x2")
        # the two numbers side by side: `value` is the bf16 headline BASELINE.json names; `at_tolerance` is
        # the fastest measured configuration of THIS run whose complete-step rel-L2 against the CPU oracle is below
        # north_star's 1e-3 (normally the fp16 build with the split trunk; the run's own line if it already meets it)
        cands = [("this run", line.get("parity"), line["value"], line["ms_per_step"])]
        for key in ("fp16_build", "fp16_split_trunk_build"):
            leg = line.get(key)
            if leg:
                cands.append((key, leg.get("parity"), leg["value"], leg["ms_per_step"]))
        ok = [c for c in cands if c[1] and c[1].get("meets_north_star") and c[1].get("ok")]
        if ok:
            best = max(ok, key=lambda c: c[2])
            line["at_tolerance"] = {"value": best[2], "unit": "steps/s", "ms_per_step": best[3], "rel_l2": best[1]["rel_l2"],
                                    "tolerance": NORTH_STAR_TOL, "configuration": best[0],
                                    "vs_headline_ms": round(best[3] / line["ms_per_step"], 4)}
        else:
            line["at_tolerance"] = None
    print(json.dumps(line))


if __name__ == "__main__":
    main()
