#!/usr/bin/env python
"""What would an fp32 (or two-term bf16) RESIDUAL TRUNK under bf16 branch activations buy?  

CPU only, oracle only (test infrastructure; it lives under tests/ because only tests/, smoke() and the
benchmark's checker leg may import the oracle): the oracle's `store()` marks sit exactly where the HIP path writes an activation
to HBM; the marks of the residual stream (block inputs / outputs, every tensor a branch result is added back into) are
tagged `trunk=True`.  This script runs the ControlNet -> UNet pair of the oracle three times on the same weights / inputs --
  fp32                    no storage rounding (the reference),
  bf16 everywhere         every store rounded to bf16 (what the HIP path does; this realisation's error ~ the HIP error),
  fp32 trunk              trunk stores kept in fp32, every branch store (norm outputs, conv1 / q / k / v / attention / GEGLU
                          intermediates, ControlNet residual outputs across the API) rounded to bf16,
  fp16 everywhere / fp32 trunk + fp16 branches   the same two with fp16 (the reference's own autocast dtype,
                          config/a100l.yaml:9) -- what the fp16 activation-storage build of the
                          inference kernels realises,
and prints the model-level relative L2 errors and the largest |value| that passes any storage point (fp16 overflows at
65 504).  usage: python tests/trunk_precision_study.py [--full]
(--full: SVD widths at 2 frames x 32 x 32, ~1 minute; default: tiny config)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ctrlv_ref as R  # noqa: E402
from tests.parity_utils import make_inputs, oracle_forward, rel_l2  # noqa: E402


def build(cfg, seed=0):
    ou = R.UNetSpatioTemporalConditionModel(time_context_order="sb", **cfg)
    R.seeded_init_(ou, seed)
    oc = R.ControlNetModel.from_unet(ou, time_context_order="sb")
    R.seeded_init_(oc, seed + 1, zero_conv_std=0.02)
    for m in (ou, oc):
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(p.to(torch.bfloat16).float())
        m.eval()
    return ou, oc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    cases = [("tiny config, B=2 F=3 16x16", dict(R.TINY_CONFIG), (2, 3, 16, 16))]
    if a.full:
        cases.append(("SVD widths, B=1 F=2 32x32", dict(R.SVD_CONFIG), (1, 2, 32, 32)))
    for name, cfg, (B, F, h, w) in cases:
        ou, oc = build(cfg)
        inp = make_inputs(cfg, B, F, h, w)
        ref = oracle_forward(ou, oc, inp, with_unet_no_ctrl=False)
        with R.storage_rounding(torch.bfloat16):
            all16 = oracle_forward(ou, oc, inp, with_unet_no_ctrl=False)
        with R.storage_rounding(torch.bfloat16, trunk_dtype=None):
            trunk32 = oracle_forward(ou, oc, inp, with_unet_no_ctrl=False)
        with R.storage_rounding(torch.float16), R.store_absmax() as amax:
            h16 = oracle_forward(ou, oc, inp, with_unet_no_ctrl=False)
        with R.storage_rounding(torch.float16, trunk_dtype=None):
            h16t = oracle_forward(ou, oc, inp, with_unet_no_ctrl=False)
        # split-plane trunk formats under fp16 branches: hi = rne_fp16(v) plus a lo plane rne(v - hi) in fp16 (what
        # trunk_dtype="fp16x2" stores: 4 bytes per element) or in 8-bit e5m2 (the top byte of that fp16 lo: 3 bytes)
        def split(lo_dtype):
            def f(x):
                hi = x.to(torch.float16).float()
                return hi + (x - hi).to(lo_dtype).float()
            return f
        with R.storage_rounding(torch.float16, trunk_dtype=split(torch.float16)):
            h16x2 = oracle_forward(ou, oc, inp, with_unet_no_ctrl=False)
        with R.storage_rounding(torch.float16, trunk_dtype=split(torch.float8_e5m2)):
            h16x8 = oracle_forward(ou, oc, inp, with_unet_no_ctrl=False)
        print(f"{name}")
        for key in ("unet", "mid"):
            print(f"  {key:5s}  bf16 everywhere {rel_l2(all16[key], ref[key]):.3e}   fp32 trunk + bf16 branches "
                  f"{rel_l2(trunk32[key], ref[key]):.3e}   fp16 everywhere {rel_l2(h16[key], ref[key]):.3e}   "
                  f"fp32 trunk + fp16 branches {rel_l2(h16t[key], ref[key]):.3e}")
        for key in ("unet", "mid"):
            print(f"  {key:5s}  fp16 branches over a split trunk: hi + fp16 lo {rel_l2(h16x2[key], ref[key]):.3e}   "
                  f"hi + e5m2 lo (3 bytes per element) {rel_l2(h16x8[key], ref[key]):.3e}")
        srt = sorted(amax)
        print(f"  max |value| over the {len(amax)} storage points of the fp16 run: {srt[-1]:.1f} (median {srt[len(srt) // 2]:.2f}, "
              f"smallest per-tensor max {srt[0]:.3g}); fp16 max 65504")


if __name__ == "__main__":
    main()
