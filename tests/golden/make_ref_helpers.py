#!/usr/bin/env python
"""Golden vectors produced BY THE REFERENCE'S OWN CODE (build container only: reads /root/reference, which does not exist on
the GPU box).  The reference holds no tests or fixtures and its model arithmetic lives in diffusers (not installable here);
the only reference-held executable code on or next to the hot path is

  * the pure-torch CLIP pre-processing helpers `_resize_with_antialiasing` / `_gaussian_blur2d` / `_filter2d` /
    `_gaussian` / `_compute_padding` (src/ctrlv/bbox_generator_baseline/utils/image_encoder.py:184-291 -- the same
    functions the pipelines call before the CLIP image encoder, pipeline_video_control.py:214-221), and
  * the EDM pre-conditioning / loss statements of the training step (tools/train_video_controlnet.py:410, 468-478).

This script extracts exactly those definitions / statements from the reference files with `ast` (no reference text is
written into this repository), executes them here on seeded inputs and stores inputs + outputs in
tests/golden/ref_helpers.npz.  tests/test_ref_helpers.py (CPU) compares ctrlv_amd's own implementations with them.
usage: python tests/golden/make_ref_helpers.py"""
import ast
import os

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _functions(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"torch": torch}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)      # the reference's own definition
    missing = [n for n in names if n not in ns]
    assert not missing, missing
    return ns


def _assignments(path, targets):
    """The reference's assignment statements to `targets` (first occurrence each, in file order), compiled as written."""
    tree = ast.parse(open(path).read())
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            n = node.targets[0].id
            if n in targets:
                found.setdefault(n, []).append(node)
    return found


def main():
    out = {}
    # ---- resize with anti-aliasing
    ns = _functions(os.path.join(REF, "src/ctrlv/bbox_generator_baseline/utils/image_encoder.py"),
                    ["_resize_with_antialiasing", "_compute_padding", "_filter2d", "_gaussian", "_gaussian_blur2d"])
    g = torch.Generator().manual_seed(2024)
    # (small images, the same down-scale FACTORS as the pipelines' 320x512 / 576x1024 -> 224x224: 1.4 / 2.3 and 2.6 / 4.6,
    # plus an odd size and a mixed up / down case; the factor sets sigma and the kernel size)
    for i, (shape, size) in enumerate([((2, 3, 80, 128), (56, 56)), ((1, 3, 144, 256), (56, 56)),
                                       ((1, 3, 97, 131), (64, 48)), ((1, 3, 50, 56), (56, 56))]):
        x = torch.rand(shape, generator=g) * 2 - 1
        y = ns["_resize_with_antialiasing"](x, size)
        out[f"resize{i}_in"], out[f"resize{i}_out"] = x.numpy(), y.numpy()
        out[f"resize{i}_size"] = np.array(size)
    # ---- EDM pre-conditioning + loss (statements of the training loop, executed in file order)
    path = os.path.join(REF, "tools/train_video_controlnet.py")
    st = _assignments(path, {"inp_noisy_latents", "c_out", "c_skip", "denoised_latents", "weighting", "loss"})
    B, F, h, w = 2, 3, 8, 12
    latents, noise = torch.randn(B, F, 4, h, w, generator=g), torch.randn(B, F, 4, h, w, generator=g)
    sig = torch.tensor([0.7, 5.3])
    sigmas = sig.reshape(B, 1, 1, 1, 1)
    ns = {"torch": torch, "sigmas": sigmas, "noisy_latents": latents + noise * sigmas, "target_latents": latents,
          "model_pred": torch.randn(B, F, 4, h, w, generator=g)}
    loss_nodes = [n for n in st["loss"] if "mean" in ast.unparse(n)][:2]      # `loss = torch.mean(...)`, `loss = loss.mean()`
    for node in [st["inp_noisy_latents"][0], st["c_out"][0], st["c_skip"][0], st["denoised_latents"][0], st["weighting"][0]] + loss_nodes:
        exec(compile(ast.Module([node], []), path, "exec"), ns)
    for k in ("noisy_latents", "target_latents", "model_pred", "inp_noisy_latents", "denoised_latents", "loss"):
        out["edm_" + k] = ns[k].numpy()
    out["edm_sigmas"] = sig.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_helpers.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_helpers.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
