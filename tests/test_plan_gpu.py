"""The plan-level C ABI (include/ctrlv_hip.h: ctrlv_plan_*, ctrlv_unet_forward, ctrlv_controlnet_forward).

* The C++ walk over the layer list (csrc/plan.hip) and the per-op Python executor (models/blocks.py) issue the same
  kernels with the same descriptors: their outputs must be BIT-IDENTICAL -- for the UNet with and without ControlNet
  residuals, the ControlNet, both temporal-context orders, CFG and single batches, bf16 and fp32 samples, and with the
  weights handed over as device bf16 or as host fp32 tensors (the packing runs in C++ on the device either way).
* `test_plan_from_a_foreign_host`: a forward driven with nothing but ctypes and raw pointers -- config struct, tensor
  descriptors by diffusers key name, caller-owned workspace / outputs -- the way a non-Python host would bind it
  (INTEGRATION.md section 2).
Model-level parity of the plan path against the oracle is what tests/test_models_gpu.py and tests/test_fullwidth_gpu.py
measure (the plan is the models' default executor).
"""
import ctypes

import pytest
import torch

from tests.parity_utils import make_inputs, make_pair

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fwd(hu, hc, inputs, dtype, with_ctrl=True):
    sample, t, ehs, ids, cond = inputs
    d = lambda x: x.to(device=DEV, dtype=dtype)   # noqa: E731
    down = mid = None
    outs = {}
    if with_ctrl:
        down, mid = hc(d(sample), t.to(DEV), d(ehs), ids.to(DEV), control_cond=d(cond), conditioning_scale=0.7,
                       return_dict=False)
        outs["down"] = [x.clone() for x in down]
        outs["mid"] = mid.clone()
    outs["unet"] = hu(d(sample), t.to(DEV), d(ehs), ids.to(DEV), down, mid).sample.clone()
    torch.cuda.synchronize()
    return outs


@torch.no_grad()
@pytest.mark.parametrize("order", ["sb", "bs"])
@pytest.mark.parametrize("B,F,h,w", [(2, 3, 16, 16), (1, 5, 24, 8)])
@pytest.mark.parametrize("mdt,dtype", [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32),
                                       (torch.float16, torch.float16), (torch.float16, torch.float32)])
def test_plan_and_python_executors_are_bit_identical(hip_lib, order, B, F, h, w, mdt, dtype):
    """mdt = the models' dtype (bf16: libctrlv_hip.so, fp16: libctrlv_hip_f16.so), dtype = the sample's."""
    import ctrlv_ref as R
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, DEV, time_context_order=order, dtype=mdt)
    inputs = make_inputs(cfg, B, F, h, w, dtype=mdt)
    res = {}
    for ex in ("plan", "python"):
        hu.executor = hc.executor = ex
        res[ex] = _fwd(hu, hc, inputs, dtype)
        res[ex + "_plain"] = _fwd(hu, hc, inputs, dtype, with_ctrl=False)
    assert hu._plan is not None and hc._plan is not None and hu._packed and hc._packed     # both executors really ran
    assert hu._plan.dtype == mdt and hu._pk["cin_w"].dtype == mdt and res["plan"]["mid"].dtype == dtype
    a, b = res["plan"], res["python"]
    assert a["unet"].dtype == dtype and torch.equal(a["unet"], b["unet"])
    assert torch.equal(a["mid"], b["mid"]) and all(torch.equal(x, y) for x, y in zip(a["down"], b["down"]))
    assert [tuple(x.shape) for x in a["down"]] == [tuple(x.shape) for x in b["down"]]
    assert torch.equal(res["plan_plain"]["unet"], res["python_plain"]["unet"])
    assert not torch.equal(a["unet"], res["plan_plain"]["unet"])                           # residuals matter


@torch.no_grad()
@pytest.mark.parametrize("order", ["sb", "bs"])
@pytest.mark.parametrize("B,F,h,w", [(2, 3, 16, 16), (1, 5, 24, 8)])
def test_plan_and_python_executors_are_bit_identical_with_a_split_trunk(hip_lib, order, B, F, h, w):
    """trunk_dtype = "fp16x2" (the residual trunk as hi + lo fp16 planes) in BOTH executors: the per-op Python
    executor carries the lo plane of every trunk tensor beside its hi plane (Workspace.trunk) and hands the same operands to
    the same launches as csrc/plan.hip -- ControlNet residuals, UNet output with and without them, bit for bit; and the mode
    really is on (the plain fp16 trunk gives other bits)."""
    import ctrlv_ref as R
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, DEV, time_context_order=order, dtype=torch.float16)
    inputs = make_inputs(cfg, B, F, h, w, dtype=torch.float16)
    res = {}
    for m in (hu, hc):
        m.trunk_dtype = "fp16x2"
    for ex in ("plan", "python"):
        hu.executor = hc.executor = ex
        res[ex] = _fwd(hu, hc, inputs, torch.float16)
        res[ex + "_plain"] = _fwd(hu, hc, inputs, torch.float16, with_ctrl=False)
    a, b = res["plan"], res["python"]
    assert torch.equal(a["unet"], b["unet"]) and torch.equal(a["mid"], b["mid"])
    assert all(torch.equal(x, y) for x, y in zip(a["down"], b["down"]))
    assert torch.equal(res["plan_plain"]["unet"], res["python_plain"]["unet"])
    for m in (hu, hc):
        m.trunk_dtype = "same"
    same = _fwd(hu, hc, inputs, torch.float16)
    assert not torch.equal(same["unet"], b["unet"])
    hu.executor = hc.executor = "plan"


@torch.no_grad()
def test_plan_follows_parameter_updates_and_context_order(hip_lib):
    """load_state_dict / .to() rebuild the plan; flipping time_context_order on a live model reaches the plan."""
    import ctrlv_ref as R
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, DEV)
    inputs = make_inputs(cfg, 2, 3, 16, 16)
    y0 = _fwd(hu, hc, inputs, torch.bfloat16)["unet"]
    hu.time_context_order = hc.time_context_order = "bs"
    y1 = _fwd(hu, hc, inputs, torch.bfloat16)["unet"]
    assert not torch.equal(y0, y1)
    hu.executor = hc.executor = "python"
    assert torch.equal(_fwd(hu, hc, inputs, torch.bfloat16)["unet"], y1)
    hu.executor = hc.executor = "plan"
    sd = {k: v.clone() for k, v in hu.state_dict().items()}
    sd["conv_out.bias"] += 1.0
    hu.load_state_dict(sd)
    assert hu._plan is None
    y2 = _fwd(hu, hc, inputs, torch.bfloat16)["unet"]
    assert (y2.float() - y1.float() - 1.0).abs().max() < 0.05


@torch.no_grad()
def test_plan_from_a_foreign_host(hip_lib):
    """Everything through ctypes: what cgo / JNI / N-API code would do."""
    import ctrlv_ref as R
    from ctrlv_amd import _lib
    from ctrlv_amd.plan import config_struct
    lib = _lib.load()
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, DEV)
    B, F, h, w = 2, 3, 16, 16
    sample, t, ehs, ids, cond = make_inputs(cfg, B, F, h, w)
    ref = _fwd(hu, hc, (sample, t, ehs, ids, cond), torch.bfloat16)

    def make_plan(kind, model):
        c = config_struct(kind, model.config, "sb")
        hnd = ctypes.c_void_p()
        assert lib.ctrlv_plan_create(ctypes.byref(c), 0, ctypes.byref(hnd)) == 0
        sd = {k: v.detach().float().cpu().contiguous() for k, v in model.state_dict().items()}     # HOST fp32 tensors
        arr = (_lib.TensorDesc * len(sd))()
        for i, (k, v) in enumerate(sd.items()):
            arr[i].name, arr[i].data, arr[i].dtype, arr[i].on_device, arr[i].numel = k.encode(), v.data_ptr(), 0, 0, v.numel()
        rc = lib.ctrlv_plan_load_weights(hnd, arr, len(sd))
        assert rc == 0, _lib.last_error()
        return hnd, sd

    pc, _keep_c = make_plan("controlnet", hc)
    pu, _keep_u = make_plan("unet", hu)
    n = lib.ctrlv_plan_num_down_residuals(pc)
    assert n == 12 == lib.ctrlv_plan_num_down_residuals(pu)
    bf = torch.bfloat16
    s_d, c_d, e_d = sample.to(DEV, bf), cond.to(DEV, bf), ehs.to(DEV, bf)
    t_d, i_d = t.reshape(1).to(DEV, torch.float32), ids.to(DEV, torch.float32)
    rows = []
    for i in range(n + 1):
        m, c = ctypes.c_int64(), ctypes.c_int32()
        assert lib.ctrlv_plan_residual_shape(pc, i, B, F, h, w, ctypes.byref(m), ctypes.byref(c)) == 0
        rows.append(torch.empty(m.value, c.value, dtype=bf, device=DEV))
    assert rows[0].shape == (B * F * h * w, 64) and rows[-1].shape == (B * F * (h // 8) * (w // 8), 128)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws_c = torch.empty(lib.ctrlv_plan_workspace_bytes(pc, B, F, h, w), dtype=torch.uint8, device=DEV)
    ws_u = torch.empty(lib.ctrlv_plan_workspace_bytes(pu, B, F, h, w), dtype=torch.uint8, device=DEV)
    assert ws_c.numel() > 0 and ws_u.numel() > 0
    outs = (ctypes.c_void_p * n)(*[r.data_ptr() for r in rows[:n]])
    rc = lib.ctrlv_controlnet_forward(pc, s_d.data_ptr(), c_d.data_ptr(), 2, t_d.data_ptr(), 1, e_d.data_ptr(),
                                      i_d.data_ptr(), 3, 0.7, outs, rows[n].data_ptr(), B, F, h, w, ws_c.data_ptr(),
                                      ws_c.numel(), st)
    assert rc == 0, _lib.last_error()
    out = torch.empty(B, F, 4, h, w, dtype=bf, device=DEV)
    rc = lib.ctrlv_unet_forward(pu, s_d.data_ptr(), 2, t_d.data_ptr(), 1, e_d.data_ptr(), i_d.data_ptr(), 3, outs,
                                rows[n].data_ptr(), None, out.data_ptr(), B, F, h, w, ws_u.data_ptr(), ws_u.numel(), st)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    assert torch.equal(out, ref["unet"])               # host-fp32 weights round to the same bf16 the models hold
    assert torch.equal(rows[n].view(B * F, h // 8, w // 8, 128).permute(0, 3, 1, 2), ref["mid"])
    # error convention: status codes + message, never a crash
    small = torch.empty(1 << 16, dtype=torch.uint8, device=DEV)
    rc = lib.ctrlv_unet_forward(pu, s_d.data_ptr(), 2, t_d.data_ptr(), 1, e_d.data_ptr(), i_d.data_ptr(), 3, None, None,
                                None, out.data_ptr(), B, F, h, w, small.data_ptr(), small.numel(), st)
    assert rc == -5 and "workspace too small" in _lib.last_error()      # CTRLV_E_WORKSPACE, refused before any launch
    rc = lib.ctrlv_unet_forward(pu, s_d.data_ptr(), 7, t_d.data_ptr(), 1, e_d.data_ptr(), i_d.data_ptr(), 3, None, None,
                                None, out.data_ptr(), B, F, h, w, ws_u.data_ptr(), ws_u.numel(), st)
    assert rc == -4 and "dtype" in _lib.last_error()                    # CTRLV_E_BAD_DTYPE
    rc = lib.ctrlv_unet_forward(pc, s_d.data_ptr(), 2, t_d.data_ptr(), 1, e_d.data_ptr(), i_d.data_ptr(), 3, None, None,
                                None, out.data_ptr(), B, F, h, w, ws_u.data_ptr(), ws_u.numel(), st)
    assert rc == -1 and "ControlNet" in _lib.last_error()
    rc = lib.ctrlv_unet_forward(pu, s_d.data_ptr(), 2, t_d.data_ptr(), 1, e_d.data_ptr(), i_d.data_ptr(), 3, None, None,
                                None, out.data_ptr(), B, F, 12, w, ws_u.data_ptr(), ws_u.numel(), st)
    assert rc == -2 and "divisible by 8" in _lib.last_error()
    torch.cuda.synchronize()
    assert lib.ctrlv_plan_destroy(pc) == 0 and lib.ctrlv_plan_destroy(pu) == 0


@torch.no_grad()
def test_unet_encoder_forward_matches_the_python_executor(hip_lib):
    """ctrlv_unet_encoder_forward (the frozen UNet's down + mid path of the training step) hands back exactly the skip
    tensors / mid output the per-op Python executor computes (bit for bit), in the shapes of ctrlv_plan_residual_shape."""
    import ctrlv_ref as R
    from tests.parity_utils import make_inputs, make_pair
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, _ = make_pair(cfg, DEV)
    B, F, h, w = 2, 3, 16, 24
    sample, t, ehs, ids, _ = make_inputs(cfg, B, F, h, w)
    sample = sample.to(DEV, torch.bfloat16)
    ehs, ids, t = ehs.to(DEV, torch.bfloat16), ids.to(DEV), t.to(DEV)
    N = B * F
    ws = hu._ensure_ready(sample)
    ctx = hu._context(ws, sample, t, ehs, ids)
    x = hu._input_rows(ws, [sample.reshape(N, -1, h, w)], N, h, w)
    x, H, W, taps = hu._run_down_mid(ctx, x, h, w)
    want = [tp[0].clone() for tp in taps] + [x.clone()]
    plan = hu._ensure_plan(sample)
    t32, ehs_p, ids32 = hu._plan_inputs(sample, t, ehs, ids)
    shapes = [plan.residual_shape(i, B, F, h, w) for i in range(plan.n_down + 1)]
    rows = [torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=DEV) for M, C in shapes]
    plan.unet_encoder_forward(sample.contiguous(), t32, ehs_p, ids32, rows[:-1], rows[-1])
    torch.cuda.synchronize()
    assert len(rows) == len(want)
    for got, ref in zip(rows, want):
        assert got.shape == ref.shape and torch.equal(got, ref)


@torch.no_grad()
def test_plan_profile_matches_the_python_executors_accounting(hip_lib):
    """ctrlv_plan_profile: the plan brackets its OWN launches with HIP events (what bench.py's roofline leg reads).  Per
    family, the number of launches and the algorithmic FLOPs / bytes equal what the per-op Python executor records for the
    same forward (ctrlv_amd/profiler.KernelTimer) -- the two executors issue the same kernels -- and every launch has a time."""
    import ctrlv_ref as R
    from ctrlv_amd import profiler
    cfg = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(cfg, DEV)
    inputs = make_inputs(cfg, 2, 3, 16, 16)
    _fwd(hu, hc, inputs, torch.bfloat16)                      # plans exist
    hu.executor = hc.executor = "python"                      # ... and the Python executor's per-model caches (its frame
    _fwd(hu, hc, inputs, torch.bfloat16)                      # embeddings are two GEMMs per transformer on first use; the
    hu.executor = hc.executor = "plan"                        # plan computes them when the weights are loaded)
    with profiler.PlanTimer(hu, hc) as pt:
        a = _fwd(hu, hc, inputs, torch.bfloat16)
    with profiler.KernelTimer() as kt:                        # (switches the models to the Python executor)
        b = _fwd(hu, hc, inputs, torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(a["unet"], b["unet"])
    ps, ks = pt.summary(), kt.summary()
    assert set(ps) == set(ks) and len(pt.launches) > 300
    for fam in ks:
        assert ps[fam]["calls"] == ks[fam]["calls"], (fam, ps[fam], ks[fam])
        assert abs(ps[fam]["flops"] - ks[fam]["flops"]) <= 1e-9 * max(1.0, ks[fam]["flops"]), fam
        assert abs(ps[fam]["bytes"] - ks[fam]["bytes"]) <= 1e-9 * max(1.0, ks[fam]["bytes"]), fam
        assert ps[fam]["ms"] > 0.0
    with profiler.PlanTimer(hu, hc) as pt2:                   # a second use starts from an empty list
        _fwd(hu, hc, inputs, torch.bfloat16, with_ctrl=False)
    assert 0 < len(pt2.launches) < len(pt.launches)
    assert hu._plan.profile_read() == []                      # ... and leaves none behind
