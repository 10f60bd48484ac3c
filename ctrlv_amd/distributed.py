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


def shard_clips(num_clips, rank, world):
    """Round-robin clip -> rank assignment: rank r owns clips {i : i mod world == r}."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, num_clips, world))


def clip_generator(seed, clip_index, device="cpu"):
    """Per-clip RNG so that results do not depend on the number of ranks."""
    return torch.Generator(device=device).manual_seed(int(seed) + int(clip_index))


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
