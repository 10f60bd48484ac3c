"""world_size-2 gloo test of the multi-GPU path's logic: clips shard by rank with no data-path collective, per-clip
RNG makes results independent of the number of ranks, timing is reduced with MAX over ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sample_clip(clip, generator, steps=6):
    """One clip's sampling loop with the product's host logic -- ctrlv_amd.schedulers.EulerDiscreteScheduler (Karras
    schedule, scale_model_input, v-prediction Euler step), per-frame guidance, CFG combine
    (pipeline_video_control.py:287-332) -- around a deterministic stand-in for the two HIP model forwards, which cannot
    run without a GPU."""
    from ctrlv_amd.schedulers import EulerDiscreteScheduler
    sched = EulerDiscreteScheduler()
    sched.set_timesteps(steps)
    F = 3
    latents = torch.randn(1, F, 4, 8, 8, generator=generator) * sched.init_noise_sigma
    guidance = torch.linspace(1.0, 3.0, F)[None, :, None, None, None]
    for t in sched.timesteps:
        x = sched.scale_model_input(torch.cat([latents] * 2), t)
        v = torch.tanh(0.3 * x) * torch.tensor([0.5, 1.0])[:, None, None, None, None]      # (uncond, cond) "model"
        v_u, v_c = v.chunk(2)
        latents = sched.step(v_u + guidance * (v_c - v_u), t, latents).prev_sample
    return latents


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ctrlv_amd import distributed as D
    r, w, _ = D.init("gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo" and D.world_size() == world and D.rank() == rank
    res, elapsed = D.run_clips(7, _sample_clip, seed=1234)          # the product's batch-shard driver
    t = D.max_over_ranks(1.0 + rank)                                # slowest rank defines the elapsed time
    n = D.sum_over_ranks(len(res))
    out.put((rank, sorted(res), {c: v.numpy().tobytes() for c, v in res.items()}, t, n, elapsed))
    dist.destroy_process_group()


def test_clip_sharding_over_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda g: g[0])
    assert got[0][1] == [0, 2, 4, 6] and got[1][1] == [1, 3, 5]
    assert all(g[3] == 2.0 for g in got) and all(g[4] == 7 for g in got)
    assert got[0][5] == got[1][5] > 0                               # both ranks report the same (max) elapsed time
    merged = {**got[0][2], **got[1][2]}
    from ctrlv_amd import distributed as D
    single, _ = D.run_clips(7, _sample_clip, seed=1234, rank=0, world=1)   # what a 1-rank run computes
    assert sorted(merged) == list(range(7))
    assert all(merged[c] == single[c].numpy().tobytes() for c in range(7))   # bit-identical per clip, any world size
    assert len({merged[c] for c in range(7)}) == 7                          # ... and the clips differ from each other


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a torchrun environment starts the two rank processes itself (the parent makes
    no GPU call) and relays rank 0's JSON; `n_gpus` is the world size the process group reports.  Launcher self-test
    mode (`--launcher-selftest`: gloo, no models) -- the only part of the N > 1 path that can run without GPUs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--launcher-selftest"], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["launcher_selftest"] is True and out["value"] is None
    assert out["ranks_seen"] == [0, 1]
    # a torchrun-style environment whose world size disagrees with --gpus is an error, not a silent 1-GPU run
    env1 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                       capture_output=True, text=True, timeout=120, env=env1)
    assert r.returncode != 0 and "process group has 1 ranks" in (r.stderr + r.stdout)


def test_shard_helpers_single_process():
    from ctrlv_amd import distributed as D
    assert D.shard_clips(8, 3, 8) == [3] and D.shard_clips(3, 5, 8) == []
    assert sorted(sum((D.shard_clips(10, r, 4) for r in range(4)), [])) == list(range(10))
    assert D.max_over_ranks(3.5) == 3.5 and D.sum_over_ranks(2) == 2.0
    import pytest
    with pytest.raises(ValueError):
        D.shard_clips(4, 4, 4)


def _grad_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ctrlv_amd import distributed as D
    from ctrlv_amd.training import allreduce_gradients
    D.init("gloo")
    g = torch.Generator().manual_seed(7)
    params = [torch.nn.Parameter(torch.randn(n, generator=g)) for n in (1000, 3, 4096, 17, 2500)]
    for i, p in enumerate(params):
        if not (i == 1 and rank == 1):                    # rank 1 has no gradient for parameter 1 (unused branch)
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    n_buckets = allreduce_gradients(params, bucket_bytes=8192)          # 1000 | 3 | 4096 | 17 + 2500 floats ...
    out.put((rank, n_buckets, [float(p.grad[0]) for p in params], [bool((p.grad == p.grad[0]).all()) for p in params]))
    dist.destroy_process_group()


def test_gradient_allreduce_buckets_over_gloo_world2():
    """The training step's data-parallel reduction (ctrlv_amd.training.allreduce_gradients; reference: DDP under
    accelerate, train_video_controlnet.py:225,485): flat fp32 buckets, all launched before the first wait, averaged;
    a parameter without a gradient on one rank contributes zeros."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n_buckets, first, uniform in got:
        assert n_buckets >= 3 and all(uniform)
        # mean over ranks of (rank + 1) * (i + 1) = 1.5 * (i + 1); parameter 1: (2 + 0) / 2
        assert first == [1.5, 1.0, 4.5, 6.0, 7.5]


def _overlap_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ctrlv_amd import distributed as D
    from ctrlv_amd.training import GradientBuckets
    D.init("gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(),
                              torch.nn.Linear(256, 8))
    unused = torch.nn.Parameter(torch.ones(5))                # never reaches the loss: no gradient on any rank
    params = list(net.parameters()) + [unused]
    gb = GradientBuckets(params, bucket_bytes=64 * 1024)
    res = []
    for step in range(2):                                     # the second step checks that finish() re-arms the hooks
        x = torch.randn(16, 64, generator=torch.Generator().manual_seed(100 * step + rank))
        for p in params:
            p.grad = None
        net(x).square().mean().backward()
        launched_in_backward = list(gb.launch_order)          # buckets whose all-reduce started before backward returned
        local = [p.grad.clone() for p in net.parameters()]
        n = gb.finish()
        res.append((n, launched_in_backward, [g.numpy().tobytes() for g in local],
                    [p.grad.numpy().tobytes() for p in net.parameters()], float(unused.grad.abs().max())))
    # accumulation (no_sync): two micro-batches add up locally without any collective, the third reduces the SUM
    for p in params:
        p.grad = None
    gb.enabled = False
    for mb in range(2):
        net(torch.randn(4, 64, generator=torch.Generator().manual_seed(7 + mb + 10 * rank))).square().mean().backward()
    assert gb.launch_order == [] and gb.finish() == 0
    gb.enabled = True
    net(torch.randn(4, 64, generator=torch.Generator().manual_seed(9 + 10 * rank))).square().mean().backward()
    local_sum = [p.grad.clone() for p in net.parameters()]       # (hooks fired on the accumulated .grad)
    gb.finish()
    res.append((local_sum[0].numpy().tobytes(), next(net.parameters()).grad.numpy().tobytes()))
    # a parameter learnt as unused (no gradient in the steps above) that receives one after all, on ONE rank only and late
    # (it is the first layer's companion: its hook fires after its bucket has gone out): nothing may be dropped
    late = unused
    for p in params:
        p.grad = None
    x = torch.randn(16, 64, generator=torch.Generator().manual_seed(500 + rank))
    y = net(x).square().mean()
    if rank == 1:
        y = y + (late * torch.arange(5.0)).sum()
    y.backward()
    gb.finish()
    res.append((late.grad.numpy().tobytes(), next(net.parameters()).grad.numpy().tobytes()))
    out.put((rank, res))
    dist.destroy_process_group()


def test_gradient_buckets_overlap_backward_over_gloo_world2():
    """GradientBuckets: buckets are all-reduced from post-accumulate-grad hooks in the order autograd completes them (last
    layer first), i.e. during the backward pass; the result is the mean of the two ranks' local gradients, identical on
    both ranks; a parameter without a gradient is flushed as zeros in finish()."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    (l0, w0), (l1, w1) = got[0][3], got[1][3]            # the late gradient of rank 1: mean of (0, arange(5)), on both ranks
    assert l0 == l1 and w0 == w1
    assert np.allclose(np.frombuffer(l0, np.float32), np.arange(5.0) / 2)
    (s0, a0), (s1, a1) = got[0][2], got[1][2]            # accumulated micro-batches: mean over ranks of the local SUMS
    assert a0 == a1
    assert np.allclose(np.frombuffer(a0, np.float32), (np.frombuffer(s0, np.float32) + np.frombuffer(s1, np.float32)) / 2,
                       rtol=1e-6, atol=1e-7)
    for step in range(2):
        (n0, early0, loc0, avg0, u0), (n1, early1, loc1, avg1, u1) = got[0][step], got[1][step]
        assert n0 == n1 >= 3 and u0 == u1 == 0.0
        # launched inside backward, in BUCKET ORDER (the collective order is fixed).  Step 0 does not know yet that the
        # parameter at the head of bucket 0 never gets a gradient: it waits for it, so everything goes out in finish();
        # from step 1 on the unused parameter is known and all buckets but (at most) the last go out during backward.
        assert early0 == early1 and early0 == list(range(len(early0)))
        assert len(early0) >= (n0 - 1 if step == 1 else 0)
        assert avg0 == avg1
        for a, b, m in zip(loc0, loc1, avg0):
            mean = (np.frombuffer(a, np.float32) + np.frombuffer(b, np.float32)) / 2
            assert np.allclose(np.frombuffer(m, np.float32), mean, rtol=1e-6, atol=1e-7)
