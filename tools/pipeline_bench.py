#!/usr/bin/env python
"""End-to-end clip latency of `StableVideoControlPipeline.__call__` (pipeline_video_control.py:105-360) at the reference's
evaluation settings (576x1024, 25 frames, 25 steps, decode_chunk_size 8, tools/eval_video_controlnet.py:79-89): encode the
conditioning image and the 25 bbox frames (VAE encoder, PyTorch-ROCm), 25 denoising steps (HIP graph), decode (VAE temporal
decoder on the HIP kernels).  Random-init full-size UNet / ControlNet / VAE, a stand-in CLIP (tests/fakes.py) -- the CLIP
image encoder is one 224x224 ViT-H forward per clip.  Prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dev = "cuda:0"
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
    import bench
    from ctrlv_amd.models import AutoencoderKLTemporalDecoder
    from ctrlv_amd.pipelines import StableVideoControlPipeline
    from ctrlv_amd.schedulers import EulerDiscreteScheduler
    from tests.fakes import FakeCLIP, fake_feature_extractor
    unet, ctrl = bench.build_models(torch.device(dev), "box2video", 25, torch.bfloat16)
    vae = AutoencoderKLTemporalDecoder().to(dev, torch.bfloat16).eval()
    clip = FakeCLIP(1024).to(dev, torch.bfloat16)
    pipe = StableVideoControlPipeline(vae, clip, unet, ctrl, EulerDiscreteScheduler(), fake_feature_extractor)
    pipe.set_progress_bar_config(disable=True)
    g = torch.Generator().manual_seed(1)
    image = (torch.rand(1, 3, 576, 1024, generator=g) * 2 - 1).to(dev, torch.bfloat16)
    cond = (torch.rand(1, 25, 3, 576, 1024, generator=g) * 2 - 1).to(dev)
    times = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.no_grad():
            fr = pipe(image, cond_images=cond, height=576, width=1024, num_frames=25, num_inference_steps=25,
                      decode_chunk_size=8, output_type="pt", generator=torch.Generator().manual_seed(rep)).frames
        torch.cuda.synchronize()
        times.append(time.time() - t0)
    ok = bool(torch.isfinite(fr.float()).all())
    print(json.dumps({"metric": "clip latency, StableVideoControlPipeline.__call__ 576x1024 x 25 frames x 25 steps",
                      "seconds_per_clip": round(min(times[1:]), 3), "all_runs_s": [round(t, 3) for t in times],
                      "first_call_includes": "HIP-graph capture, MIOpen find (fast mode), weight packing",
                      "frames": list(fr.shape), "finite": ok, "vae_decode": os.environ.get("CTRLV_VAE_HIP", "1") != "0" and "hip" or "torch"}))


if __name__ == "__main__":
    main()
