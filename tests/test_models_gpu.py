"""GPU parity of the HIP UNet / ControlNet against the CPU oracle (same seeded bf16-rounded weights, same inputs).

Three comparisons per case (tests/parity_utils.run_parity), every bound rel-L2 AND element-wise (`parity_err`):

  storage  HIP vs the fp32 oracle with bf16 rounding at exactly the HIP path's activation-storage points
           (ctrlv_ref.storage_rounding).  This is the arithmetic check: what remains is accumulation
           order and flipped roundings that re-amplify through 55 blocks.  Bound TOL_STORAGE.
  fp32     HIP vs the plain fp32 oracle.  Dominated by bf16 activation storage (each block contributes 2-3e-3, see
           tests/test_oracle.py::test_golden_block_vectors_bf16_weights_and_storage_rounding).  Bound TOL_FP32.
  yardstick  the same network executed by PyTorch itself in bf16 on the GPU (rocBLAS / MIOpen / SDPA) vs the fp32
           oracle: HIP's error must not exceed it by more than 10 %.

  storage  (see above) at MODEL level is a second bf16 realisation of the network, not a tighter reference: measured at
           full width, HIP-vs-storage (1.0-1.5e-2) ~ sqrt(HIP-vs-fp32^2 + storage-vs-fp32^2) -- the two roundings
           decorrelate after a few blocks (a flipped bf16 rounding re-amplifies through GroupNorm / attention), i.e. the
           model-level error is rounding noise, not bias.  Per block, where no amplification has happened yet, the
           storage-rounded oracle IS the tight reference (tests/test_blocks_gpu.py: 2e-3).  Bound here: 2e-2.
  trace    per-block error growth (parity_utils.error_growth_trace): the rel-L2 error after every res block /
           transformer in execution order must stay below TOL_FP32 and no single block may add more than TOL_STEP --
           a wiring or kernel defect is a jump, storage noise a slow walk.

north_star's "1e-3 relative bf16 tolerance" holds per kernel (tests/test_ops_gpu.py: 1e-4 with fp32 output) and per
block against the storage-rounded oracle (tests/test_blocks_gpu.py: 2e-3); through the whole network bf16 storage
itself costs ~1e-2 whoever executes it, which the yardstick demonstrates.
"""
import pytest
import torch

from tests.parity_utils import make_inputs, make_pair, run_parity

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_FP32 = 1.5e-2
TOL_STORAGE = 2e-2
TOL_STEP = 5e-3


def check_tables(err):
    assert max(err["storage"].values()) < TOL_STORAGE, err
    assert max(err["fp32"].values()) < TOL_FP32, err
    for k, v in err["torch_bf16"].items():
        assert err["fp32_l2"][k] <= 1.1 * v + 1e-4, (k, err["fp32_l2"][k], v)


def check_trace(trace):
    prev = 0.0
    for name, e in trace:
        print(f"  {e:.2e}  {name}")
    for name, e in trace:
        assert e < TOL_FP32, (name, e)
        assert e - prev < TOL_STEP, (name, e, prev)
        prev = max(prev, e)


def test_tiny_error_growth_trace(hip_lib):
    import ctrlv_ref as R
    from tests.parity_utils import error_growth_trace
    cfg = dict(R.TINY_CONFIG)
    ou, oc, hu, hc = make_pair(cfg, DEV)
    tr = error_growth_trace(ou, hu, make_inputs(cfg, 2, 3, 16, 16), DEV, oc, hc)
    assert len(tr) == 17 + 38                       # ControlNet: 10 res + 7 transformers; UNet: 22 + 16
    check_trace(tr)


@pytest.mark.parametrize("order", ["sb", "bs"])
def test_tiny_unet_controlnet_parity(hip_lib, order):
    import ctrlv_ref as R
    check_tables(run_parity(dict(R.TINY_CONFIG), DEV, B=2, F=3, h=16, w=16, time_context_order=order, verbose=True))


TOL_F16 = 3e-3          # fp16 element build (libctrlv_hip_f16.so): north_star's 1e-3 with head-room for the element bound


@pytest.mark.parametrize("order", ["sb", "bs"])
def test_tiny_unet_controlnet_parity_fp16(hip_lib, order):
    """The fp16 activation-storage build (the reference's own autocast dtype, config/a100l.yaml:9) against the fp32
    oracle: rel-L2 AND element bound below 3e-3 at model level (measured ~1.3e-3: profiles/r04_storage_precision_study.txt
    predicts exactly that from the oracle alone), against 1.5e-2 for bf16 storage."""
    import ctrlv_ref as R
    err = run_parity(dict(R.TINY_CONFIG), DEV, B=2, F=3, h=16, w=16, time_context_order=order, verbose=True,
                     torch_bf16=False, dtype=torch.float16)
    assert max(err["fp32"].values()) < TOL_F16 and max(err["storage"].values()) < TOL_F16, err


def test_tiny_parity_batch1_ragged_fp16(hip_lib):
    import ctrlv_ref as R
    err = run_parity(dict(R.TINY_CONFIG), DEV, B=1, F=5, h=24, w=8, verbose=True, torch_bf16=False, dtype=torch.float16)
    assert max(err["fp32"].values()) < TOL_F16, err


def test_tiny_error_growth_trace_fp16(hip_lib):
    import ctrlv_ref as R
    from tests.parity_utils import error_growth_trace
    cfg = dict(R.TINY_CONFIG)
    ou, oc, hu, hc = make_pair(cfg, DEV, dtype=torch.float16)
    tr = error_growth_trace(ou, hu, make_inputs(cfg, 2, 3, 16, 16, dtype=torch.float16), DEV, oc, hc)
    assert len(tr) == 17 + 38
    for name, e in tr:
        print(f"  {e:.2e}  {name}")
    assert max(e for _, e in tr) < TOL_F16, tr


@torch.no_grad()
def test_fp16_models_are_deterministic_and_zero_controlnet_is_a_noop(hip_lib):
    import ctrlv_ref as R
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, DEV, zero_conv_std=None, dtype=torch.float16)
    sample, t, ehs, ids, cond = make_inputs(cfg, 2, 3, 16, 16, dtype=torch.float16)
    d = lambda x: x.to(device=DEV, dtype=torch.float16)   # noqa: E731
    down, mid = hc(d(sample), t.to(DEV), d(ehs), ids.to(DEV), control_cond=d(cond), return_dict=False)
    assert down[0].dtype == torch.float16
    assert all(x.abs().max().item() == 0 for x in down) and mid.abs().max().item() == 0
    y0 = hu(d(sample), t.to(DEV), d(ehs), ids.to(DEV)).sample
    y1 = hu(d(sample), t.to(DEV), d(ehs), ids.to(DEV), down, mid).sample
    y2 = hu(d(sample), t.to(DEV), d(ehs), ids.to(DEV)).sample
    assert y0.dtype == torch.float16 and torch.equal(y0, y1) and torch.equal(y0, y2)


def test_tiny_parity_batch1_ragged(hip_lib):
    import ctrlv_ref as R
    check_tables(run_parity(dict(R.TINY_CONFIG), DEV, B=1, F=5, h=24, w=8, verbose=True))


@torch.no_grad()
def test_zero_controlnet_is_noop_and_deterministic(hip_lib):
    """Structural known-answers: zero-initialised zero-convs => residuals are exactly 0 and the UNet
    output is bit-identical with and without them; two runs are bit-identical (no atomics anywhere)."""
    import ctrlv_ref as R
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, DEV, zero_conv_std=None)
    sample, t, ehs, ids, cond = make_inputs(cfg, 2, 3, 16, 16)
    d = lambda x: x.to(device=DEV, dtype=torch.bfloat16)   # noqa: E731
    down, mid = hc(d(sample), t.to(DEV), d(ehs), ids.to(DEV), control_cond=d(cond), return_dict=False)
    assert all(x.abs().max().item() == 0 for x in down) and mid.abs().max().item() == 0
    y0 = hu(d(sample), t.to(DEV), d(ehs), ids.to(DEV)).sample
    y1 = hu(d(sample), t.to(DEV), d(ehs), ids.to(DEV), down, mid).sample
    y2 = hu(d(sample), t.to(DEV), d(ehs), ids.to(DEV)).sample
    assert torch.equal(y0, y1) and torch.equal(y0, y2)


@torch.no_grad()
def test_clips_are_independent(hip_lib):
    """Batch-shard property behind the multi-GPU path: with the upstream-fixed context order a 2-clip forward equals
    the two 1-clip forwards bit-for-bit (no cross-clip arithmetic anywhere in the path)."""
    import ctrlv_ref as R
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, DEV, time_context_order="bs")
    sample, t, ehs, ids, cond = make_inputs(cfg, 2, 3, 16, 16)
    d = lambda x: x.to(device=DEV, dtype=torch.bfloat16)   # noqa: E731
    down, mid = hc(d(sample), t.to(DEV), d(ehs), ids.to(DEV), control_cond=d(cond), return_dict=False)
    y = hu(d(sample), t.to(DEV), d(ehs), ids.to(DEV), down, mid).sample.clone()
    for b in range(2):
        s = slice(b, b + 1)
        dn, md = hc(d(sample[s]), t.to(DEV), d(ehs[s]), ids[s].to(DEV), control_cond=d(cond[s]), return_dict=False)
        yb = hu(d(sample[s]), t.to(DEV), d(ehs[s]), ids[s].to(DEV), dn, md).sample
        assert torch.equal(yb[0], y[b])


@torch.no_grad()
def test_foreign_nchw_residuals_and_dtypes(hip_lib):
    """Residuals handed over as plain contiguous NCHW fp32 tensors (what a diffusers ControlNet would return) give the
    same result as the channels-last bf16 views of ctrlv_amd's own ControlNet; fp32 in -> fp32 out."""
    import ctrlv_ref as R
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, DEV)
    sample, t, ehs, ids, cond = make_inputs(cfg, 1, 3, 16, 16)
    d = lambda x: x.to(device=DEV, dtype=torch.bfloat16)   # noqa: E731
    down, mid = hc(d(sample), t.to(DEV), d(ehs), ids.to(DEV), control_cond=d(cond), return_dict=False)
    y = hu(d(sample), t.to(DEV), d(ehs), ids.to(DEV), down, mid).sample
    down_f = [x.float().contiguous() for x in down]
    y2 = hu(sample.to(DEV), t.to(DEV), ehs.to(DEV), ids.to(DEV), down_f, mid.float().contiguous()).sample
    assert y2.dtype == torch.float32
    assert torch.equal(y2.to(torch.bfloat16), y)


def test_input_validation(hip_lib):
    import ctrlv_ref as R
    from ctrlv_amd._lib import CtrlvHipError
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, DEV)
    sample, t, ehs, ids, cond = make_inputs(cfg, 1, 3, 16, 16)
    d = lambda x: x.to(device=DEV, dtype=torch.bfloat16)   # noqa: E731
    with pytest.raises(ValueError):      # latent size not divisible by 8
        hu(d(sample[..., :12, :]), t, d(ehs), ids.to(DEV))
    with pytest.raises(ValueError):      # more than one encoder token
        hu(d(sample), t, d(ehs.repeat(1, 2, 1)), ids.to(DEV))
    with pytest.raises(ValueError):      # control_cond missing
        hc(d(sample), t, d(ehs), ids.to(DEV))
    with pytest.raises(CtrlvHipError):   # CPU tensors: no fallback path
        hu(sample.to(torch.bfloat16), t, ehs.to(torch.bfloat16), ids)
