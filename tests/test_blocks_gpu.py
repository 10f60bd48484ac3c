"""Per-block GPU parity against the committed golden fixtures (tests/golden/block_vectors.npz, `q*` keys).

SpatioTemporalResBlock (with 1x1 shortcut; with a skip-concat input as on the up path) and
TransformerSpatioTemporalModel (batch 2, both temporal-context orders -- SURVEY H1) run standalone through the HIP
path on the fixtures' inputs and bf16-rounded weights, and are compared with
  * `*_y`  : the oracle's plain fp32 result  -> tolerance 6e-3 (one block's worth of bf16 activation storage:
             the fixture pair itself differs by 2.2e-3 .. 3.2e-3, asserted in tests/test_oracle.py), and
  * `*_ys` : the oracle with bf16 rounding at exactly the HIP path's storage points -> tolerance 2e-3
             (north_star's "1e-3 relative bf16 tolerance" regime: what is left is accumulation order and the
             occasional flipped bf16 rounding, no systematic term).
Both bounds are rel-L2 AND element-wise (tests/parity_utils.parity_err).
The per-clip tables the model normally prepares once per forward (time_emb_proj(silu(emb)), to_out(to_v(clip))) are
built here with the same HIP GEMMs.
"""
import importlib.util
import os

import numpy as np
import pytest
import torch

from tests.parity_utils import parity_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
B, F, H, W = 2, 3, 8, 8
TOL_FP32, TOL_STORAGE = 6e-3, 2e-3


@pytest.fixture(scope="module")
def fx(hip_lib):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    with torch.no_grad():
        blocks = mg.q_blocks()
    gold = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "block_vectors.npz")).items()}
    return blocks, gold


def rows(x):                       # (N, C, H, W) fp32 -> channels-last bf16 rows on the device
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous().to(DEV, torch.bfloat16)


def nchw(r, n, h, w):
    return r.float().cpu().reshape(n, h, w, -1).permute(0, 3, 1, 2)


def make_ctx(temb_b=None, res_block=None, ehs_b=None, transformer=None, order="sb"):
    """FwdCtx with the per-clip row-vector tables of one block, computed by the HIP GEMMs (as encoder._context does
    for the whole model)."""
    from ctrlv_amd import ops, packing
    from ctrlv_amd.models.blocks import FwdCtx
    from ctrlv_amd.workspace import Workspace
    temb = xattn = None
    if res_block is not None:
        lins = res_block.temb_projections()
        wcat = torch.cat([l.weight.detach().float() for l in lins], 0)
        bcat = torch.cat([l.bias.detach().float() for l in lins], 0)
        emb = temb_b.to(DEV, torch.bfloat16)
        emb_s = torch.empty_like(emb)
        ops.silu(emb, emb_s)
        temb = torch.empty(emb.shape[0], wcat.shape[0], dtype=torch.float32, device=DEV)
        ops.gemm(emb_s, packing.pack_linear(wcat).to(DEV), temb, N=wcat.shape[0], cin=emb.shape[1],
                 bias=bcat.to(DEV), out_f32=True)
        res_block.temb_off = (0, lins[0].weight.shape[0])
    if transformer is not None:
        C = transformer.C
        attns = transformer.cross_attentions()
        wv = torch.cat([a.to_v.weight.detach().float() for a in attns], 0)
        e = ehs_b.reshape(ehs_b.shape[0], -1).to(DEV, torch.bfloat16).contiguous()
        v_all = torch.empty(e.shape[0], 2 * C, dtype=torch.bfloat16, device=DEV)
        ops.gemm(e, packing.pack_linear(wv).to(DEV), v_all, N=2 * C, cin=e.shape[1])
        xattn = torch.empty(e.shape[0], 2 * C, dtype=torch.float32, device=DEV)
        for i, a in enumerate(attns):
            ops.gemm(v_all[:, i * C:(i + 1) * C], packing.pack_linear(a.to_out[0].weight.detach().float()).to(DEV),
                     xattn[:, i * C:(i + 1) * C], N=C, cin=C, bias=a.to_out[0].bias.detach().float().to(DEV),
                     out_f32=True)
        transformer.xattn_off = (0, C)
    return FwdCtx(Workspace(DEV), B, F, temb, xattn, order)


def hip_block(cls, oracle_block, *args, **kw):
    blk = cls(*args, **kw)
    missing, unexpected = blk.load_state_dict(oracle_block.state_dict(), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    blk = blk.to(DEV, torch.bfloat16)
    blk.pack()
    return blk


def check(got, gold, key):
    e32 = parity_err(got, gold[key + "_y"], key + " vs fp32 oracle")
    es = parity_err(got, gold[key + "_ys"], key + " vs bf16-storage oracle")
    assert e32 < TOL_FP32 and es < TOL_STORAGE, (key, e32, es)


@torch.no_grad()
def test_res_block_with_shortcut_matches_golden(fx):
    from ctrlv_amd.models.blocks import SpatioTemporalResBlock
    (rb, _, _), gold = fx
    blk = hip_block(SpatioTemporalResBlock, rb, 64, 128, 256, eps=1e-6)
    ctx = make_ctx(temb_b=gold["q_temb"], res_block=blk)
    y = blk.run(ctx, rows(gold["q_x"]), H, W)
    torch.cuda.synchronize()
    check(nchw(y, B * F, H, W), gold, "qres")


@torch.no_grad()
def test_res_block_on_skip_concat_matches_golden(fx):
    """Up-path form: torch.cat([hidden, skip], dim=1) is never materialised -- GroupNorm, conv1 and the shortcut read
    the two sources in place."""
    from ctrlv_amd.models.blocks import SpatioTemporalResBlock
    (_, rc, _), gold = fx
    blk = hip_block(SpatioTemporalResBlock, rc, 64 + 128, 64, 256, eps=1e-5)
    ctx = make_ctx(temb_b=gold["q_temb"], res_block=blk)
    y = blk.run(ctx, rows(gold["q_x"]), H, W, x2=rows(gold["q_skip"]))
    torch.cuda.synchronize()
    check(nchw(y, B * F, H, W), gold, "qcat")


@torch.no_grad()
@pytest.mark.parametrize("order", ["sb", "bs"])
def test_transformer_matches_golden(fx, order):
    from ctrlv_amd.models.blocks import TransformerSpatioTemporalModel
    (_, _, tr), gold = fx
    blk = hip_block(TransformerSpatioTemporalModel, tr, 2, 64, 128, 64)
    ctx = make_ctx(ehs_b=gold["q_ehs"], transformer=blk, order=order)
    y = blk.run(ctx, rows(gold["q_xt"]), H, W)
    torch.cuda.synchronize()
    check(nchw(y, B * F, H, W), gold, "qtr_" + order)
    other = "bs" if order == "sb" else "sb"
    assert parity_err(nchw(y, B * F, H, W), gold[f"qtr_{other}_y"]) > 2e-2      # the two orders are distinguishable
