#!/usr/bin/env python
"""What comes after the loop: time of the once-per-clip VAE temporal decode of a 25-frame 576x1024 clip on
PyTorch-ROCm (ctrlv_amd.models.AutoencoderKLTemporalDecoder, random-init weights, reference chunking of 8 / 14 frames,
pipeline_video_control.py:346), next to the 25-step denoising loop it follows.  Informational: decides when the decoder
deserves HIP kernels of its own."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ctrlv_amd.models import AutoencoderKLTemporalDecoder
    dev = "cuda:0"
    dt = torch.float16 if "--fp16" in sys.argv else torch.bfloat16
    vae = AutoencoderKLTemporalDecoder().to(dev, dt).eval()
    lat = torch.randn(25, 4, 72, 128, device=dev, dtype=dt)
    if "--encode" in sys.argv:          # the once-per-clip encode of the 25 bbox frames (pipeline_video_control.py:71-101)
        img = torch.randn(25, 3, 576, 1024, device=dev, dtype=dt)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            with torch.no_grad():
                out = [vae.encode(img).latent_dist.mode()]           # all 25 frames at once, as the pipeline does
            torch.cuda.synchronize()
            print(f"  encode pass {rep}: {time.time() - t0:.2f} s for 25 frames, latents {tuple(torch.cat(out).shape)}", flush=True)
        return
    for chunk in ([int(a) for a in sys.argv[1:] if a.isdigit()] or [8, 14, 25]):
        for rep in range(2):            # first pass: MIOpen kernel search
            torch.cuda.synchronize()
            t0 = time.time()
            with torch.no_grad():
                out = [vae.decode(lat[i:i + chunk], num_frames=lat[i:i + chunk].shape[0]).sample for i in range(0, 25, chunk)]
            torch.cuda.synchronize()
            dtm = time.time() - t0
            print(f"  pass {rep}: {dtm:.1f} s", flush=True)
        print(f"decode_chunk_size {chunk:2d}: {dtm * 1e3:8.1f} ms per clip ({dt}), output {tuple(torch.cat(out).shape)}, "
              f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
