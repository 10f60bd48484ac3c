"""N > 1 on real hardware (RCCL over xGMI): runs only where the box has at least two GPUs (the round's 1-GPU boxes skip
it; the gloo world-2 tests of tests/test_distributed_cpu.py cover the same logic on the CPU).

* `bench.py --gpus 2` through its own launcher: n_gpus == 2, a finite value, weak scaling;
* clip sharding (BASELINE config 4): per-clip sampling results of a 2-rank `run_clips` are bit-identical to the 1-rank run;
* data-parallel training step (config 5): GradientBuckets over RCCL gives the mean of the ranks' micro-batch gradients --
  compared with a 1-rank run that computes both micro-batches and averages -- and overlaps the backward pass.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs")


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}


@needs2
def test_bench_two_gpus(hip_lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2",
                        "--no-cpu-baseline", "--height", "256", "--width", "256", "--frames", "5"],
                       capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["finite"] and out["value"] > 0
    assert "clip-shard x2" in out["config"]["parallelism"]


@needs2
def test_two_rank_clips_and_gradients_match_one_rank(hip_lib, tmp_path):
    worker = os.path.join(ROOT, "tests", "multigpu_worker.py")
    d2, d1 = tmp_path / "w2", tmp_path / "w1"
    d2.mkdir(); d1.mkdir()
    launcher = ("import sys; sys.path.insert(0, %r); from ctrlv_amd.distributed import launch_local_ranks; "
                "launch_local_ranks(%r, [%r], 2)" % (ROOT, worker, str(d2)))
    r = subprocess.run([sys.executable, "-c", launcher], capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, worker, str(d1), "--emulate-world", "2"], capture_output=True, text=True, timeout=900,
                       env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = torch.load(d2 / "rank0.pt"), torch.load(d2 / "rank1.pt")
    one = torch.load(d1 / "rank0.pt")
    assert a["world"] == b["world"] == 2 and one["world"] == 1
    assert sorted(a["clips"]) == [0, 2] and sorted(b["clips"]) == [1, 3] and sorted(one["clips"]) == [0, 1, 2, 3]
    merged = {**a["clips"], **b["clips"]}
    for c in range(4):
        assert torch.equal(merged[c], one["clips"][c]), c          # a clip's result does not depend on the world size
    assert a["elapsed"] == b["elapsed"] > 0
    from tests.parity_utils import rel_l2
    for n, g in one["grads"].items():
        assert torch.equal(a["grads"][n], b["grads"][n]), n        # every rank holds the same averaged gradient
        if float(g.abs().max()) > 0:
            assert rel_l2(a["grads"][n], g) < 1e-5, n               # = mean of the micro-batch gradients (fp32 sum order)
    assert a["loss"][1] >= 1                                        # buckets went out DURING the backward pass
