"""Whole-model parity at PRODUCTION WIDTHS (SVD_CONFIG: channels 320/640/1280/1280, heads 5/10/20/20, cross-attention
dim 1024, 1280-wide time embedding) against the CPU oracle, at BASELINE config 1's size (2 frames, 32x32 latent).

The tiny configuration never reaches the 320-wide GEMM tile, K = 1280, the C/32 = 10 GroupNorm groups that straddle
8-column lane chunks, the fused [sum Cout, 1280] time_emb_proj GEMM with its 64 column offsets or the fused
[sum C, 1024] cross-attention to_v GEMM; this file does, with the same three comparisons and tolerances as
tests/test_models_gpu.py.  One seeded 1.5 G + 0.7 G parameter pair is built once per module (fp32 oracle on the host,
bf16 HIP models on the device); an oracle forward at this size is ~0.8 TFLOP (2-4 s on the GPU box's host cores).

Cases: B = 1 (no CFG), and B = 2 (CFG layout: zero unconditional half) under both temporal-context orders (SURVEY H1;
B = 1 cannot tell them apart).
"""
import pytest
import torch

from tests.parity_utils import make_pair, run_parity
from tests.test_models_gpu import check_tables, check_trace

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def full_pair(hip_lib):
    import ctrlv_ref as R
    cfg = dict(R.SVD_CONFIG, num_frames=2)
    return cfg, make_pair(cfg, DEV, lean=True)


@pytest.mark.parametrize("B,order", [(1, "sb"), (2, "sb"), (2, "bs")])
def test_fullwidth_unet_controlnet_parity_cfg1_size(full_pair, B, order):
    cfg, pair = full_pair
    err = run_parity(cfg, DEV, B=B, F=2, h=32, w=32, time_context_order=order, verbose=True, pair=pair,
                     with_unet_no_ctrl=(B == 1))
    check_tables(err)


@torch.no_grad()
def test_fullwidth_ragged_latent(full_pair):
    """A latent whose levels are not multiples of any tile (24 x 40 -> 12x20 -> 6x10 -> 3x5), 3 frames, CFG batch."""
    cfg, pair = full_pair
    err = run_parity(cfg, DEV, B=2, F=3, h=24, w=40, time_context_order="sb", verbose=True, pair=pair,
                     torch_bf16=False, with_unet_no_ctrl=False)
    assert max(err["storage"].values()) < 2e-2 and max(err["fp32"].values()) < 1.5e-2, err


def test_fullwidth_reference_default_latent_40x64(full_pair):
    """The reference's own default working size, 320x512 (latent 40 x 64 -> 20x32 -> 10x16 -> 5x8; SURVEY finding 5,
    src/ctrlv/datasets/kitti_abstract.py:86-90), at production widths against the oracle (2 frames, CFG batch)."""
    cfg, pair = full_pair
    err = run_parity(cfg, DEV, B=2, F=2, h=40, w=64, time_context_order="sb", verbose=True, pair=pair,
                     with_unet_no_ctrl=False)
    check_tables(err)


def test_fullwidth_error_growth_trace(full_pair):
    """Error after each of the 55 blocks at production widths (CFG batch 2): slow walk, no jump."""
    from tests.parity_utils import error_growth_trace, make_inputs, set_context_order
    cfg, (ou, oc, hu, hc) = full_pair
    set_context_order((ou, oc, hu, hc), "sb")
    tr = error_growth_trace(ou, hu, make_inputs(cfg, 2, 2, 32, 32), DEV, oc, hc)
    assert len(tr) == 55
    check_trace(tr)
