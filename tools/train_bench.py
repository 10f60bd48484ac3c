#!/usr/bin/env python
"""cfg5 on one MI355X: the ControlNet training step at the reference's size (B = 1, F = 25, 576x1024 -> 72x128 latent,
full SVD width; tools/train_video_controlnet.py:366-488) through ctrlv_amd.training.train_step.  Prints ONE JSON line:
ms per optimisation step, its split (forward / backward / optimizer, HIP events), peak device memory, and the analytic
work of SURVEY.md 8 a11 (219 TFLOP: ControlNet fwd 29.3 + UNet fwd 80.0 + ControlNet bwd 58.6 + UNet decoder dgrad 51).
Synthetic data, random-init weights (there are no checkpoints in this image).
usage: python tools/train_bench.py [--steps 3] [--warmup 1] [--frames 25] [--height 576] [--width 1024]
Data parallel (one process per GPU, RCCL): `python tools/train_bench.py --gpus N` starts the N ranks itself from a GPU-free
parent; or python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
tools/train_bench.py --gpus N ...  -- every rank trains on its own clip, the fp32 gradients are
all-reduced in 25 MB buckets launched from autograd hooks while the backward runs (ctrlv_amd.training.GradientBuckets);
the time is the MAX over ranks, `samples_per_s` the whole-job rate."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--lr", type=float, default=1e-5)
    ap.add_argument("--fused-adamw", type=int, default=1, help="torch.optim.AdamW(fused=...)")
    ap.add_argument("--gradient-checkpointing", type=int, default=0,
                    help="1: unet.enable_gradient_checkpointing() + controlnet.enable_gradient_checkpointing() (the GEGLU "
                         "feed-forward intermediates are recomputed in the backward; tools/train_video_controlnet.py:185-186)")
    ap.add_argument("--mark-steps", type=int, default=0,
                    help="1: a torch.cuda._sleep kernel in front of every step (tools/trace_last_step.py cuts a rocprofv3 "
                         "kernel trace at these marks)")
    ap.add_argument("--gpus", type=int, default=0,
                    help="N > 1 without a torchrun environment: start N rank processes (one per GPU) from this GPU-free parent")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        from ctrlv_amd.distributed import launch_local_ranks
        launch_local_ranks(__file__, sys.argv[1:], args.gpus)
        return
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    if args.gpus and args.gpus != world:
        raise SystemExit(f"train_bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from ctrlv_amd import training
    from ctrlv_amd.models import ControlNetModel, UNetSpatioTemporalConditionModel
    from ctrlv_amd.utils import build_on_device, random_init_
    unet = build_on_device(UNetSpatioTemporalConditionModel, dev, num_frames=args.frames)
    random_init_(unet, seed=0)
    ctrl = build_on_device(ControlNetModel, dev, dtype=torch.float32, num_frames=args.frames)      # fp32 masters
    random_init_(ctrl, seed=1, zero_conv_std=0.02)
    for p in unet.parameters():
        p.requires_grad_(False)
    if args.gradient_checkpointing:
        unet.enable_gradient_checkpointing()
        ctrl.enable_gradient_checkpointing()
    params = [p for p in ctrl.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=args.lr, weight_decay=1e-2, fused=bool(args.fused_adamw))
    buckets = training.GradientBuckets(params) if world > 1 else None
    B, F, h, w = 1, args.frames, args.height // 8, args.width // 8
    g = torch.Generator(device=dev).manual_seed(1234 + rank)             # every rank: its own clip
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)      # noqa: E731
    batch = dict(latents=rn(B, F, 4, h, w), noise=rn(B, F, 4, h, w), sigmas=torch.tensor([1.5], device=dev),
                 image_latents=rn(B, 1, 4, h, w).repeat(1, F, 1, 1, 1), control_cond=rn(B, F, 4, h, w),
                 encoder_hidden_states=rn(B, 1, 1024), added_time_ids=torch.tensor([[6.0, 127.0, 0.02]], device=dev))
    ev = lambda: torch.cuda.Event(enable_timing=True)             # noqa: E731
    times, losses = [], []
    for i in range(args.warmup + args.steps):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.time()
        e = [ev() for _ in range(4)]
        if args.mark_steps:
            torch.cuda._sleep(2000)
        e[0].record()
        lat, noise, sig = batch["latents"], batch["noise"], batch["sigmas"]
        s5 = sig.reshape(B, 1, 1, 1, 1)
        noisy = lat + noise * s5
        sample = torch.cat([noisy / (s5 * s5 + 1) ** 0.5, batch["image_latents"]], dim=2).to(torch.bfloat16)
        ts = 0.25 * torch.log(sig)
        down, mid = training.controlnet_train_forward(ctrl, sample, ts, batch["encoder_hidden_states"],
                                                      batch["added_time_ids"], batch["control_cond"].to(torch.bfloat16))
        pred = training.unet_train_forward(unet, sample, ts, batch["encoder_hidden_states"], batch["added_time_ids"],
                                           down, mid)
        loss = training.edm_loss(pred, noisy, lat, sig)
        e[1].record()
        loss.backward()
        if buckets is not None:
            buckets.finish()
        e[2].record()
        opt.step()
        opt.zero_grad(set_to_none=True)
        e[3].record()
        torch.cuda.synchronize()
        wall = (time.time() - t0) * 1e3
        if world > 1:                                             # slowest rank defines the step
            wt = torch.tensor([wall], device=dev)
            dist.all_reduce(wt, op=dist.ReduceOp.MAX)
            wall = float(wt)
        if i >= args.warmup:
            times.append((wall, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])))
        losses.append(float(loss.detach()))
        del down, mid, pred, loss
    n = len(times)
    avg = [sum(t[k] for t in times) / n for k in range(4)]
    tf = 219.0 * (F / 25.0) * (h * w) / (72 * 128)          # scales with pixels x frames (attention: quadratic; quoted for 72x128)
    if rank != 0:
        return
    print(json.dumps({"metric": "cfg5 ControlNet training step (B=1 per GPU, no CFG)", "n_gpus": world,
                      "samples_per_s": round(world * 1e3 / avg[0], 3), "scaling": "weak", "ms_per_step": round(avg[0], 1),
                      "forward_ms": round(avg[1], 1), "backward_ms": round(avg[2], 1), "optimizer_ms": round(avg[3], 1),
                      "steps": n, "warmup": args.warmup, "frames": F, "latent": [h, w], "analytic_tflop_per_step": round(tf, 1),
                      "tflops": round(tf / avg[0] * 1e3, 1), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                      "losses": [round(v, 5) for v in losses],
                      "gradient_checkpointing": bool(args.gradient_checkpointing), "dtype": "bf16 compute, fp32 master parameters + AdamW",
                      "data": "synthetic, random-init weights"}))


if __name__ == "__main__":
    main()
