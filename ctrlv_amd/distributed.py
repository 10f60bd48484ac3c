"""Multi-GPU plumbing: clips shard by rank, one process per GPU, RCCL (backend "nccl") over xGMI.

The denoising path has NO collective inside a forward (north_star): a rank runs the full sampling loop
of its own clips with replicated weights.  The only communication is a start/end barrier and one MAX all-reduce of the
per-rank wall time for the scaling curve (cf. `accelerator.wait_for_everyone()` at
/root/reference/tools/eval_video_controlnet.py:108).  On CPU (tests) the same code runs over gloo.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for a single process)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    """World size as the process group reports it (1 without a group)."""
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_clips(num_clips, rank, world):
    """Round-robin clip -> rank assignment: rank r owns clips {i : i mod world == r}."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, num_clips, world))


def clip_generator(seed, clip_index, device="cpu"):
    """Per-clip RNG so that results do not depend on the number of ranks."""
    return torch.Generator(device=device).manual_seed(int(seed) + int(clip_index))


def run_clips(num_clips, sample_fn, seed=0, rank=None, world=None, device="cpu", sync=None):
    """Batch-shard sampling driver (BASELINE config 4): this rank samples the clips it owns
    (`shard_clips`), each with its own `clip_generator(seed, clip)` so a clip's result does not depend on the number
    of ranks, with NO collective in the data path; a barrier brackets the region and the elapsed wall time is the MAX
    over ranks.

    sample_fn(clip_index, generator) -> anything (typically `pipeline(..., generator=generator).frames`).
    `sync` is called before each clock read (torch.cuda.synchronize on GPUs).
    Returns (results: {clip_index: value} for the owned clips, elapsed_seconds_max_over_ranks)."""
    import time
    if rank is None or world is None:
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    mine = shard_clips(num_clips, rank, world)
    barrier()
    if sync is not None:
        sync()
    t0 = time.perf_counter()
    results = {}
    for clip in mine:
        results[clip] = sample_fn(clip, clip_generator(seed, clip, device))
    if sync is not None:
        sync()
    barrier()
    return results, max_over_ranks(time.perf_counter() - t0)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """MAX all-reduce of a Python float (per-rank elapsed time)."""
    if not dist.is_initialized():
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not dist.is_initialized():
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def launch_local_ranks(script, argv, n, log=None):
    """Start `n` rank processes of `script` (one per GPU of this node) from a process that has made NO HIP call, relay
    rank 0's stdout and end the other ranks if one dies (by PID, never by pattern).  The children are ordinary
    subprocesses with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set -- what
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` would give them -- so `python bench.py --gpus 8` or
    `python tools/train_bench.py --gpus 8` is one command for a driver.  Exits the calling process with the first
    non-zero rank exit code."""
    import socket
    import subprocess
    import sys
    import threading
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(n))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    chunks = []
    th = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    th.start()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad:
            failed = bad[0].returncode
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    th.join(timeout=10)
    sys.stdout.write("".join(c for c in chunks if c))
    sys.stdout.flush()
    if failed is not None or any(rcs):
        if log is not None:
            log(f"rank exit codes {rcs}")
        sys.exit(failed or next(rc for rc in rcs if rc))
