"""Rank process of tests/test_multigpu_gpu.py (one per GPU; started by ctrlv_amd.distributed.launch_local_ranks or, for
the 1-rank reference run, directly).  Writes <outdir>/rank<r>.pt = {clips: {clip: latents}, grads: {name: tensor}, loss}.

Part 1: the batch-shard sampling driver (BASELINE config 4) -- `run_clips` over 4 clips with the real DenoiseStepper on a
        tiny seeded model pair: a clip's result must not depend on the number of ranks.
Part 2: the data-parallel training step (config 5) -- rank r's micro-batch (seed 100 + r of `world`), GradientBuckets over
        RCCL overlapped with the backward; a 1-rank run called with `--emulate-world W` computes the W micro-batches one
        after the other and averages, which is what the all-reduce must give."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--emulate-world", type=int, default=0)
    a = ap.parse_args()
    import ctrlv_ref as R
    from ctrlv_amd import distributed as D
    from ctrlv_amd.pipelines.pipeline_utils import DenoiseStepper
    from ctrlv_amd.schedulers import EulerDiscreteScheduler
    from ctrlv_amd.training import GradientBuckets, train_step
    from tests.parity_utils import make_pair
    from tests.test_train_gpu import _batch
    rank, world, local = D.init("nccl" if int(os.environ.get("WORLD_SIZE", 1)) > 1 else None)
    world = D.world_size()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, dev, seed=3, time_context_order="bs")      # seeded: identical on every rank

    def sample(clip, gen):
        F, h, w = 3, 16, 16
        sched = EulerDiscreteScheduler()
        sched.set_timesteps(4, device=dev)
        bf = torch.bfloat16
        lat = (torch.randn(1, F, 4, h, w, generator=gen) * sched.init_noise_sigma).to(dev)
        img = torch.randn(1, 4, h, w, generator=gen)
        il = torch.cat([torch.zeros_like(img), img]).unsqueeze(1).repeat(1, F, 1, 1, 1).to(dev, bf)
        e = torch.randn(1, 1, cfg["cross_attention_dim"], generator=gen)
        ehs = torch.cat([torch.zeros_like(e), e]).to(dev, bf)
        c = torch.randn(1, F, 4, h, w, generator=gen)
        cond = torch.cat([torch.zeros_like(c), c]).to(dev, bf)
        ids = torch.tensor([[6.0, 127.0, 0.02]] * 2, device=dev, dtype=bf)
        st = DenoiseStepper(hu, hc, sched, lat, il, ehs, ids, cond, 1.0, 3.0, 1.0, do_cfg=True, use_hip_graph=False)
        for i in range(4):
            st.step(i)
        return st.latents.cpu()

    with torch.no_grad():
        clips, elapsed = D.run_clips(4, sample, seed=11, sync=torch.cuda.synchronize)
    # ---- training step
    hc.float()
    for p in hc.parameters():
        p.requires_grad_(True)
    for p in hu.parameters():
        p.requires_grad_(False)
    params = [p for p in hc.parameters() if p.requires_grad]
    to_dev = lambda b: {k: v.to(dev) for k, v in b.items()}   # noqa: E731
    if a.emulate_world:
        acc, losses = None, []
        for r in range(a.emulate_world):
            for p in params:
                p.grad = None
            losses.append(float(train_step(hc, hu, to_dev(_batch(cfg, 1, 3, 16, 16, seed=100 + r)), conditioning_scale=0.8)))
            g = [p.grad.detach().float().clone() for p in params]
            acc = g if acc is None else [x + y for x, y in zip(acc, g)]
        grads = [x / a.emulate_world for x in acc]
        loss = losses
    else:
        gb = GradientBuckets(params, bucket_bytes=1 << 20)
        for _ in range(2):               # two steps: the second one knows the gradient-less parameters (overlap path)
            for p in params:
                p.grad = None
            loss = float(train_step(hc, hu, to_dev(_batch(cfg, 1, 3, 16, 16, seed=100 + rank)), conditioning_scale=0.8,
                                    buckets=gb))
        grads = [p.grad.detach().float().clone() for p in params]
        loss = [loss, len(gb.launch_order)]
    torch.cuda.synchronize()
    names = [n for n, p in hc.named_parameters() if p.requires_grad]
    torch.save(dict(clips=clips, grads={n: g.cpu() for n, g in zip(names, grads)}, loss=loss, world=world, elapsed=elapsed),
               os.path.join(a.outdir, f"rank{rank}.pt"))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
