"""SPLIT ("fp16x2") residual trunk of the fp16 element build (round 5; include/ctrlv_hip.h ctrlv_gemm_desc.out_lo,
ctrlv_plan_set_trunk_mode).

north_star: "outputs match the diffusers CPU reference ... within 1e-3 relative".  fp16 activation storage measures
1.28e-3 on the complete step; the oracle-only study (profiles/r04_storage_precision_study.txt) attributes most of it to the
re-rounding of the RESIDUAL TRUNK at each of its ~150 residual adds and predicts 5.9e-4 with the trunk kept at fp32
precision under fp16 branches.  Here the trunk tensors carry a second fp16 plane (hi = rne(v), lo = rne(v - hi): fp32's
bytes, 21+ significant bits); MFMA operands read the hi plane in place, residual operands and norm inputs read hi + lo.

Kernel level: every split-aware kernel against fp32 PyTorch on the SAME inputs at fp32-output accuracy (the split pair
carries ~2^-22), the ping-pong tile and the 2-stage kernel BIT-IDENTICAL in both planes (a clip's bits must not depend on
which kernel -- i.e. which batch size -- serves a layer).  Model level: tiny and production widths against the fp32 oracle
with north_star's bound: rel-L2 < 1e-3.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.parity_utils import compare, hip_forward, make_inputs, make_pair, oracle_forward, parity_err, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EL = torch.float16
NORTH_STAR_TOL = 1e-3          # BASELINE.json north_star: "within 1e-3 relative"


@pytest.fixture(autouse=True)
def _pack_in_fp16():
    from ctrlv_amd import packing
    with packing.element_dtype(EL):
        yield


@pytest.fixture(scope="module")
def ops(hip_lib):
    from ctrlv_amd import ops as o
    return o


def g(seed=0):
    return torch.Generator().manual_seed(seed)


LO = torch.float8_e5m2     # one byte per element: fp16's sign / exponent + two mantissa bits (csrc/common.h lo_t)


def split(v):
    """fp32 tensor -> (hi fp16, lo e5m2) planes the way the kernels store them."""
    hi = v.to(EL)
    lo = (v - hi.float()).to(LO)
    return hi, lo


def joined(hi, lo):
    return hi.float().cpu() + lo.cpu().float()


def lo_plane(*shape, nan=False):
    """device lo plane (optionally pre-filled with the e5m2 NaN byte: a launch must overwrite every element)"""
    t = torch.empty(*shape, dtype=LO, device=DEV)
    if nan:
        t.view(torch.uint8).fill_(0x7F)
    return t


def same(a, b):
    return torch.equal(a.view(torch.uint8), b.view(torch.uint8)) if a.dtype == LO else torch.equal(a, b)


def rows_from_nchw(x):
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()


def test_split_planes_carry_15_bits():
    v = torch.randn(4096, generator=g(1)) * 3
    hi, lo = split(v)
    assert rel_l2(hi.float() + lo.float(), v) < 2.5e-5 and rel_l2(hi.float(), v) > 1e-4


@pytest.mark.parametrize("M", [300, 2048 + 77])
@pytest.mark.parametrize("epi", ["bias", "r1", "r1v", "r1r2", "bias_a2"])
def test_gemm_split_epilogues(ops, M, epi):
    """mode 0 with every operand set the trunk's writers use; small M runs the 2-stage kernel, large M the 256x320
    ping-pong tile with the LO epilogue: same bits in both planes for the rows they share."""
    from ctrlv_amd import packing
    N, K = 320, 256
    A = torch.randn(M, K, generator=g(1)).to(EL)
    Wt = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    bias = torch.randn(N, generator=g(3))
    r1 = torch.randn(M, N, generator=g(4)) * 2
    r2 = torch.randn(M, N, generator=g(5)) * 2
    V = torch.randn(7, N, generator=g(6))
    Wp = packing.pack_linear(Wt)
    lin = A.float() @ Wp[:N].float().T + bias
    r1h, r1l = split(r1)
    r2h, r2l = split(r2)
    r1s, r2s = r1h.float() + r1l.float(), r2h.float() + r2l.float()
    vidx = (torch.arange(M) // 13) % 7
    kw = dict(N=N, cin=K, bias=bias.to(DEV))
    Ad = A.to(DEV)
    if epi == "bias":
        ref = lin
    elif epi == "r1":
        kw.update(R1=r1h.to(DEV), R1_lo=r1l.to(DEV), s1=0.5, s_acc=0.7)
        ref = 0.7 * lin + 0.5 * r1s
    elif epi == "r1v":
        kw.update(R1=r1h.to(DEV), R1_lo=r1l.to(DEV), V=V.to(DEV), vmode=1, vdiv=13, vmod=7)
        ref = lin + r1s + V[vidx]
    elif epi == "r1r2":
        kw.update(R1=r1h.to(DEV), R1_lo=r1l.to(DEV), s1=0.5, R2=r2h.to(DEV), R2_lo=r2l.to(DEV), s2=0.25, s_acc=0.5)
        ref = 0.5 * lin + 0.5 * r1s + 0.25 * r2s
    else:       # the skip-concat shortcut: A | A2 split on K
        A2 = torch.randn(M, 128, generator=g(7)).to(EL)
        kw.update(A2=A2.to(DEV), c_split=128)
        Ad = A[:, :128].contiguous().to(DEV)
        ref = torch.cat([A[:, :128], A2], 1).float() @ Wp[:N].float().T + bias
    outs = {}
    for tile in ([0, 1] if M > 1024 else [0]):
        hi = torch.full((M, N), float("nan"), dtype=EL, device=DEV)
        lo = lo_plane(M, N, nan=True)
        ops.gemm(Ad, Wp.to(DEV), hi, out_lo=lo, tile=tile, **kw)
        # the pair is the fp32 epilogue value to ~2^-15; the hi plane alone is the plain fp16 output
        assert parity_err(joined(hi, lo), ref, f"{epi} tile {tile}") < 1e-4
        # |lo| <= half an ulp of hi (+ its own rounding to three significant bits)
        assert (lo.cpu().float().abs() <= hi.float().cpu().abs() * 2.0 ** -11 * 1.125 + 2.0 ** -16).all()
        outs[tile] = (hi, lo)
    if len(outs) == 2:      # ping-pong LO epilogue == 2-stage epilogue, bit for bit, in both planes
        assert same(outs[0][0], outs[1][0]) and same(outs[0][1], outs[1][1])
    # without split operands the hi plane IS the plain fp16 output of the same launch
    if epi in ("bias", "bias_a2"):
        plain = torch.empty(M, N, dtype=EL, device=DEV)
        ops.gemm(Ad, Wp.to(DEV), plain, **kw)
        assert torch.equal(plain, outs[0][0])
    # a split R1 against the same launch with the hi plane alone: the lo plane is really read
    if epi == "r1":
        hi2 = torch.empty(M, N, dtype=EL, device=DEV)
        ops.gemm(Ad, Wp.to(DEV), hi2, **{k: v for k, v in kw.items() if k != "R1_lo"})
        assert parity_err(hi2, ref) > 5e-5


@pytest.mark.parametrize("kind,H,W,n,cin,cout", [("conv_r1", 8, 32, 6, 64, 320), ("conv_r1", 9, 16, 9, 64, 320),
                                                  ("conv_s2", 16, 32, 9, 64, 320), ("conv_up", 8, 16, 5, 64, 320),
                                                  ("temporal_r1", 8, 16, 9, 64, 320)])
def test_gemm_split_convs(ops, kind, H, W, n, cin, cout):
    """The conv writers of the trunk: conv2 (+R1; row-halo and per-tap gathers), the resampling convs ({bias}), the
    temporal conv2 (AlphaBlender: s_acc * acc + R1)."""
    from ctrlv_amd import packing
    x = torch.randn(n, cin, H, W, generator=g(1)).to(EL)
    b = torch.randn(cout, generator=g(3))
    if kind.startswith("conv"):
        wt = torch.randn(cout, cin, 3, 3, generator=g(2)) / math.sqrt(9 * cin)
        wd = packing.pack_conv3x3(wt).to(DEV)
        if kind == "conv_s2":
            Ho, Wo, stride, up = H // 2, W // 2, 2, 0
            ref = F.conv2d(x.float(), wt.to(EL).float(), b, stride=2, padding=1)
        elif kind == "conv_up":
            Ho, Wo, stride, up = 2 * H, 2 * W, 1, 1
            ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), wt.to(EL).float(), b, padding=1)
        else:
            Ho, Wo, stride, up = H, W, 1, 0
            ref = F.conv2d(x.float(), wt.to(EL).float(), b, padding=1)
        kw = dict(N=cout, cin=cin, taps=9, mode=1, conv=(H, W, Ho, Wo, stride, up), bias=b.to(DEV))
        ref = rows_from_nchw(ref)
    else:
        Ho, Wo = H, W
        wt = torch.randn(cout, cin, 3, 1, 1, generator=g(2)) / math.sqrt(3 * cin)
        wd = packing.pack_conv_temporal(wt).to(DEV)
        Fr = 3
        x5 = x.float().reshape(n // Fr, Fr, cin, H, W).permute(0, 2, 1, 3, 4)
        ref = F.conv3d(x5, wt.to(EL).float(), b, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(n, cout, H, W)
        ref = rows_from_nchw(ref)
        kw = dict(N=cout, cin=cin, taps=3, mode=2, temporal=(Fr, H * W), bias=b.to(DEV))
    M = n * Ho * Wo
    if kind.endswith("_r1"):
        r1 = torch.randn(M, cout, generator=g(4)) * 2
        r1h, r1l = split(r1)
        kw.update(R1=r1h.to(DEV), R1_lo=r1l.to(DEV), s_acc=0.5)
        ref = 0.5 * ref + (r1h.float() + r1l.float())
    xd = rows_from_nchw(x).to(DEV)
    outs = {}
    for tile in ([0, 1] if M >= 1024 else [0]):
        hi = torch.full((M, cout), float("nan"), dtype=EL, device=DEV)
        lo = lo_plane(M, cout, nan=True)
        ops.gemm(xd, wd, hi, out_lo=lo, tile=tile, **kw)
        assert parity_err(joined(hi, lo), ref, f"{kind} tile {tile}") < 1e-4
        outs[tile] = (hi, lo)
    if len(outs) == 2:
        assert same(outs[0][0], outs[1][0]) and same(outs[0][1], outs[1][1])


@pytest.mark.parametrize("kind,H,W,n,ips,cout", [("conv_r1", 8, 32, 6, 1, 320), ("conv_r1", 16, 64, 3, 3, 640),
                                                   ("temporal_r1", 8, 16, 9, 3, 320), ("temporal_r1", 8, 8, 6, 3, 1280)])
def test_gemm_split_writes_groupnorm_partials(ops, kind, H, W, n, ips, cout):
    """Round 6: the two trunk writers that feed a GroupNorm -- conv2 of a res block (row-halo 3x3, {R1})
    and its temporal conv2 (AlphaBlender, {R1}; in front of a transformer's opening norm) -- write the norm's chunk partials
    in the SAME launch that splits the output into hi + lo (GNS + LO instantiations): both planes are bit-identical to the
    launch without partials, and GroupNorm(hi + lo) from the partials agrees with fp64 and with the two-pass split norm."""
    from ctrlv_amd import packing
    cin, S, M = 64, H * W, n * H * W
    x = torch.randn(n, cin, H, W, generator=g(1)).to(EL)
    b = torch.randn(cout, generator=g(3)) + 3.0
    if kind.startswith("conv"):
        wt = torch.randn(cout, cin, 3, 3, generator=g(2)) / math.sqrt(9 * cin)
        wd = packing.pack_conv3x3(wt).to(DEV)
        kw = dict(N=cout, cin=cin, taps=9, mode=1, conv=(H, W, H, W, 1, 0), bias=b.to(DEV))
        ref = rows_from_nchw(F.conv2d(x.double(), wt.to(EL).double(), b.double(), padding=1))
    else:
        wt = torch.randn(cout, cin, 3, 1, 1, generator=g(2)) / math.sqrt(3 * cin)
        wd = packing.pack_conv_temporal(wt).to(DEV)
        x5 = x.double().reshape(n // ips, ips, cin, H, W).permute(0, 2, 1, 3, 4)
        ref = F.conv3d(x5, wt.to(EL).double(), b.double(), padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(n, cout, H, W)
        ref = rows_from_nchw(ref)
        kw = dict(N=cout, cin=cin, taps=3, mode=2, temporal=(ips, S), bias=b.to(DEV))
    r1 = torch.randn(M, cout, generator=g(4)) * 2
    r1h, r1l = split(r1)
    kw.update(R1=r1h.to(DEV), R1_lo=r1l.to(DEV), s_acc=0.5)
    ref = 0.5 * ref + (r1h.double() + r1l.double())
    xd = rows_from_nchw(x).to(DEV)
    hi0, lo0 = torch.empty(M, cout, dtype=EL, device=DEV), lo_plane(M, cout)
    ops.gemm(xd, wd, hi0, out_lo=lo0, **kw)
    hi = torch.full((M, cout), float("nan"), dtype=EL, device=DEV)
    lo = lo_plane(M, cout, nan=True)
    assert ops.gemm_gn_partials_serves(xd, wd, hi, out_lo=lo, **kw)
    part = torch.full((ops.groupnorm_fused_scratch_floats(n, S, ips),), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm(xd, wd, hi, out_lo=lo, gn_partials=part, **kw)
    assert same(hi, hi0) and same(lo, lo0)
    assert parity_err(joined(hi, lo), ref.float(), kind) < 1e-4
    gamma, beta = torch.randn(cout, generator=g(6)), torch.randn(cout, generator=g(7))
    y = torch.full((M, cout), float("nan"), dtype=EL, device=DEV)
    ops.groupnorm_from_partials(hi, n, S, cout, ips, gamma.to(DEV), beta.to(DEV), 1e-6, True, y, part, x_lo=lo)
    xs = ref.reshape(n, H, W, cout).permute(0, 3, 1, 2)
    if ips == 1:
        gn = F.group_norm(xs, 32, gamma.double(), beta.double(), 1e-6)
    else:
        x5 = xs.reshape(n // ips, ips, cout, H, W).permute(0, 2, 1, 3, 4)
        gn = F.group_norm(x5, 32, gamma.double(), beta.double(), 1e-6).permute(0, 2, 1, 3, 4).reshape(n, cout, H, W)
    yr = rows_from_nchw(F.silu(gn).float())
    assert parity_err(y, yr, f"{kind}: GroupNorm from split partials") < 5e-4
    y2 = torch.empty_like(y)
    p2 = torch.empty(ops.groupnorm_scratch_floats(n, S, cout, ips), dtype=torch.float32, device=DEV)
    ops.groupnorm(hi, None, n, S, cout, ips, gamma.to(DEV), beta.to(DEV), 1e-6, True, y2, p2, x_lo=lo)
    assert rel_l2(y.float().cpu(), y2.float().cpu()) < 2e-4
    # the {bias} / {R1, V} / {R1, R2} trunk writers feed no GroupNorm: not served with split planes
    kw2 = {k: v for k, v in kw.items() if k not in ("R1", "R1_lo", "s_acc")}
    assert not ops.gemm_gn_partials_serves(xd, wd, hi, out_lo=lo, **kw2)


def test_gemm_split_rejections(ops):
    from ctrlv_amd import packing
    M, N, K = 2048, 320, 128
    A = torch.randn(M, K, generator=g(1)).to(EL).to(DEV)
    Wp = packing.pack_linear(torch.randn(N, K, generator=g(2))).to(DEV)
    hi = torch.empty(M, N, dtype=EL, device=DEV)
    lo = lo_plane(M, N)
    with pytest.raises(ValueError):          # the 256-wide ping-pong tile has no split epilogue
        ops.gemm(A, Wp, hi, N=N, cin=K, out_lo=lo, tile=5)
    with pytest.raises(ValueError):          # a lo plane without its hi operand
        ops.gemm(A, Wp, hi, N=N, cin=K, out_lo=lo, R1_lo=lo)
    with pytest.raises(ValueError):          # SiLU epilogues are branch outputs: no split form
        ops.gemm(A, Wp, hi, N=N, cin=K, out_lo=lo, act=1)
    # the bf16 library does not serve split planes at all
    Ab, Wb = A.to(torch.bfloat16), Wp.to(torch.bfloat16)
    with pytest.raises(ValueError):
        ops.gemm(Ab, Wb, hi.to(torch.bfloat16), N=N, cin=K, out_lo=lo)


@pytest.mark.parametrize("C,H,W,n,ips", [(320, 9, 16, 6, 1), (320, 9, 16, 6, 3), (640, 8, 8, 4, 2), (320, 72, 128, 1, 1)])
def test_groupnorm_split_input(ops, C, H, W, n, ips):
    """GroupNorm(+SiLU) of x + x_lo, plain and as the skip concat (x | x2) with a lo plane on either half."""
    S = H * W
    x = torch.randn(n, C, H, W, generator=g(1)) * 1.5 + 0.7
    xr = rows_from_nchw(x)
    hi, lo = split(xr)
    gamma, beta = torch.randn(C, generator=g(2)), torch.randn(C, generator=g(3))
    xs = (hi.float() + lo.float()).reshape(n, H, W, C).permute(0, 3, 1, 2)
    if ips == 1:
        ref = F.group_norm(xs.double(), 32, gamma.double(), beta.double(), 1e-5)
    else:
        x5 = xs.double().reshape(n // ips, ips, C, H, W).permute(0, 2, 1, 3, 4)
        ref = F.group_norm(x5, 32, gamma.double(), beta.double(), 1e-5).permute(0, 2, 1, 3, 4).reshape(n, C, H, W)
    ref = rows_from_nchw(F.silu(ref).float())
    y = torch.empty(n * S, C, dtype=EL, device=DEV)
    part = torch.empty(ops.groupnorm_scratch_floats(n, S, C, ips), dtype=torch.float32, device=DEV)
    ops.groupnorm(hi.to(DEV), None, n, S, C, ips, gamma.to(DEV), beta.to(DEV), 1e-5, True, y, part, x_lo=lo.to(DEV))
    assert parity_err(y, ref) < 5e-4
    # the lo plane matters: the hi plane alone is a different (coarser) input
    y0 = torch.empty_like(y)
    ops.groupnorm(hi.to(DEV), None, n, S, C, ips, gamma.to(DEV), beta.to(DEV), 1e-5, True, y0, part)
    assert not torch.equal(y, y0)
    # concat form: channels [0, c1) from x (split), the rest from x2 (split / plain)
    c1 = C // 2 if (C // 2) % 8 == 0 else 160
    a_h, a_l = hi[:, :c1].contiguous(), lo[:, :c1].contiguous()
    b_h, b_l = hi[:, c1:].contiguous(), lo[:, c1:].contiguous()
    y2 = torch.empty_like(y)
    ops.groupnorm(a_h.to(DEV), b_h.to(DEV), n, S, C, ips, gamma.to(DEV), beta.to(DEV), 1e-5, True, y2, part, x_lo=a_l.to(DEV),
                  x2_lo=b_l.to(DEV))
    assert torch.equal(y2, y)


@pytest.mark.parametrize("C", [64, 320, 640, 1280])
def test_layernorm_split_input(ops, C):
    M = 1000
    x = torch.randn(M, C, generator=g(1)) * 2 + 0.3
    hi, lo = split(x)
    gamma, beta = torch.randn(C, generator=g(2)), torch.randn(C, generator=g(3))
    V = torch.randn(5, C, generator=g(4))
    xs = hi.float() + lo.float()
    y = torch.empty(M, C, dtype=EL, device=DEV)
    ops.layernorm(hi.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, y, x_lo=lo.to(DEV))
    assert parity_err(y, F.layer_norm(xs, (C,), gamma, beta, 1e-5)) < 5e-4
    ops.layernorm(hi.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, y, V=V.to(DEV), vdiv=100, vmod=5, x_lo=lo.to(DEV))
    ref = F.layer_norm(xs + V[(torch.arange(M) // 100) % 5], (C,), gamma, beta, 1e-5)
    assert parity_err(y, ref) < 5e-4


def test_axpby_split(ops):
    n = 8 * 12345
    x = torch.randn(n, generator=g(1)) * 3
    r = torch.randn(n, generator=g(2)).to(EL)
    hi, lo = split(x)
    yh, yl = torch.empty(n, dtype=EL, device=DEV), lo_plane(n)
    ops.axpby_split(hi.to(DEV), lo.to(DEV), r.to(DEV), 1.0, 1.0, yh, yl)
    ref = hi.float() + lo.float() + r.float()
    assert rel_l2(joined(yh, yl), ref) < 2.5e-5
    # in place, as the plan runs it
    xh, xl = hi.to(DEV), lo.to(DEV)
    ops.axpby_split(xh, xl, r.to(DEV), 1.0, 1.0, xh, xl)
    assert same(xh, yh) and same(xl, yl)


@pytest.mark.parametrize("M,epi", [(1000, "r1"), (2048 + 72, "r1r2"), (256 * 9, "r1v")])
def test_ff_fused_split(ops, M, epi):
    """The fused C = 320 feed-forward with split R1 / R2 / out against the two split launches (same arithmetic up to the
    summation order of the second projection)."""
    from ctrlv_amd import packing
    C = 320
    x = torch.randn(M, C, generator=g(1)).to(EL)
    W1 = torch.randn(8 * C, C, generator=g(2)) / math.sqrt(C)
    b1 = torch.randn(8 * C, generator=g(3)) * 0.1
    W2 = torch.randn(C, 4 * C, generator=g(4)) / math.sqrt(4 * C)
    b2 = torch.randn(C, generator=g(5)) * 0.1
    W1p, b1p = packing.pack_geglu(W1, b1)
    W2p = packing.pack_linear(W2)
    r1h, r1l = split(torch.randn(M, C, generator=g(6)) * 2)
    r2h, r2l = split(torch.randn(M, C, generator=g(7)) * 2)
    V = torch.randn(9, C, generator=g(8))
    kw = dict(R1=r1h.to(DEV), R1_lo=r1l.to(DEV))
    if epi == "r1r2":
        kw.update(R2=r2h.to(DEV), R2_lo=r2l.to(DEV), s_acc=0.5, s1=0.5, s2=0.5)
    if epi == "r1v":
        kw.update(V=V.to(DEV), vmode=1, vdiv=256, vmod=9)
    xd, W1d, b1d, W2d, b2d = x.to(DEV), W1p.to(DEV), b1p.to(DEV), W2p.to(DEV), b2.to(DEV)
    w1f, w2f = ops.ff_fused_pack(W1d, b1d, W2d)
    hi, lo = torch.full((M, C), float("nan"), dtype=EL, device=DEV), lo_plane(M, C, nan=True)
    assert ops.ff_fused_serves(xd, hi, out_lo=lo, **kw)
    ops.ff_fused(xd, w1f, w2f, hi, bias=b2d, out_lo=lo, **kw)
    u = torch.empty(M, 4 * C, dtype=EL, device=DEV)
    ops.gemm(xd, W1d, u, N=8 * C, cin=C, bias=b1d, geglu=1)
    hi2, lo2 = torch.empty_like(hi), torch.empty_like(lo)
    ops.gemm(u, W2d, hi2, N=C, cin=4 * C, bias=b2d, out_lo=lo2, **kw)
    assert parity_err(joined(hi, lo), joined(hi2, lo2)) < 2e-4      # (fp32 summation order of the second projection)
    proj = x.float() @ W1.to(EL).float().T + b1
    h = (proj[:, :4 * C] * F.gelu(proj[:, 4 * C:])).to(EL).float()
    ref = h @ W2.to(EL).float().T + b2
    r1s, r2s = r1h.float() + r1l.float(), r2h.float() + r2l.float()
    if epi == "r1":
        ref = ref + r1s
    elif epi == "r1r2":
        ref = 0.5 * ref + 0.5 * r1s + 0.5 * r2s
    else:
        ref = ref + r1s + V[(torch.arange(M) // 256) % 9]
    assert parity_err(joined(hi, lo), ref) < 6e-4        # (u is an fp16 branch tensor in both)


# ------------------------------------------------------------------------------------------------ model level
def _set_trunk(models, mode):
    for m in models:
        m.trunk_dtype = mode


@torch.no_grad()
def _model_errs(cfg, pair, B, Fr, h, w, order="sb"):
    """rel-L2 / parity_err of the HIP pair against the fp32 oracle, split trunk and plain fp16 trunk."""
    from tests.parity_utils import set_context_order
    ou, oc, hu, hc = pair
    set_context_order(pair, order)
    inputs = make_inputs(cfg, B, Fr, h, w, dtype=EL)
    ref = oracle_forward(ou, oc, inputs, with_unet_no_ctrl=False)
    out = {}
    for mode in ("fp16x2", "same"):
        _set_trunk((hu, hc), mode)
        got = hip_forward(hu, hc, inputs, DEV, with_unet_no_ctrl=False)
        assert hu._plan is not None and hu._plan.trunk_mode == mode and hc._plan.trunk_mode == mode
        out[mode] = dict(l2=compare(got, ref, rel_l2), both=compare(got, ref))
        print(f"trunk {mode}: " + "  ".join(f"{k}={v:.2e}" for k, v in out[mode]["l2"].items()))
    _set_trunk((hu, hc), "same")
    return out


def test_tiny_model_split_trunk(hip_lib):
    import ctrlv_ref as R
    cfg = dict(R.TINY_CONFIG)
    pair = make_pair(cfg, DEV, dtype=EL)
    e = _model_errs(cfg, pair, 2, 3, 16, 16)
    # (the tiny config is noisier than production widths -- narrow layers average less: the oracle-only study predicts 7.0e-4
    #  for its UNet and 8.7e-4 for its mid residual, against 5.9e-4 / 7.4e-4 at SVD widths; north_star's number is asserted on
    #  the production widths below)
    assert e["fp16x2"]["l2"]["unet"] < NORTH_STAR_TOL, e
    for k in ("unet", "controlnet_mid", "controlnet_down"):
        assert e["fp16x2"]["l2"][k] < 1.25e-3, e
        assert e["fp16x2"]["both"][k] < 1.5e-3, e
        assert e["fp16x2"]["l2"][k] < 0.8 * e["same"]["l2"][k], e       # the trunk really is the larger half of the error


def test_split_trunk_matches_the_oracles_trunk_study(hip_lib):
    """HIP with the split trunk vs the oracle with fp16 rounding at every BRANCH storage point and an fp32 trunk
    (ctrlv_ref.storage_rounding(trunk_dtype=None): the configuration of profiles/r04_storage_precision_study.txt)."""
    import ctrlv_ref as R
    cfg = dict(R.TINY_CONFIG)
    ou, oc, hu, hc = make_pair(cfg, DEV, dtype=EL)
    inputs = make_inputs(cfg, 2, 3, 16, 16, dtype=EL)
    with R.storage_rounding(EL, trunk_dtype=None):
        ref_q = oracle_forward(ou, oc, inputs, with_unet_no_ctrl=False)
    _set_trunk((hu, hc), "fp16x2")
    got = hip_forward(hu, hc, inputs, DEV, with_unet_no_ctrl=False)
    assert max(compare(got, ref_q).values()) < 1.5e-3


def test_split_trunk_clip_independence_and_determinism(hip_lib):
    """A clip's bits do not depend on the batch it is computed in (upstream-fixed "bs" context order), nor on the run."""
    import ctrlv_ref as R
    cfg = dict(R.TINY_CONFIG)
    ou, oc, hu, hc = make_pair(cfg, DEV, dtype=EL, time_context_order="bs")
    _set_trunk((hu, hc), "fp16x2")
    inputs = make_inputs(cfg, 2, 3, 16, 16, dtype=EL)
    both = hip_forward(hu, hc, inputs, DEV, with_unet_no_ctrl=False)
    again = hip_forward(hu, hc, inputs, DEV, with_unet_no_ctrl=False)
    assert torch.equal(both["unet"], again["unet"]) and torch.equal(both["mid"], again["mid"])
    for b in range(2):
        one = tuple(x[b:b + 1] if torch.is_tensor(x) and x.dim() > 0 else x for x in inputs)
        got = hip_forward(hu, hc, one, DEV, with_unet_no_ctrl=False)
        assert torch.equal(got["unet"][0], both["unet"][b])
        assert torch.equal(got["mid"], both["mid"][b * 3:(b + 1) * 3])


def test_split_trunk_needs_fp16(hip_lib):
    import ctrlv_ref as R
    cfg = dict(R.TINY_CONFIG)
    ou, oc, hu, hc = make_pair(cfg, DEV)                  # bf16 models
    hu.trunk_dtype = "fp16x2"
    inputs = make_inputs(cfg, 1, 2, 16, 16)
    sample, t, ehs, ids, cond = inputs
    with pytest.raises(ValueError):
        hu(sample.to(DEV, torch.bfloat16), t.to(DEV), ehs.to(DEV, torch.bfloat16), ids.to(DEV))
    hu.trunk_dtype = "nonsense"
    with pytest.raises(ValueError):
        hu(sample.to(DEV, torch.bfloat16), t.to(DEV), ehs.to(DEV, torch.bfloat16), ids.to(DEV))


@pytest.fixture(scope="module")
def full_pair_f16(hip_lib):
    import ctrlv_ref as R
    cfg = dict(R.SVD_CONFIG, num_frames=2)
    return cfg, make_pair(cfg, DEV, lean=True, dtype=EL)


@pytest.mark.parametrize("B,h,w", [(2, 32, 32), (1, 72, 128)])
def test_fullwidth_split_trunk_meets_north_star(full_pair_f16, B, h, w):
    """Production widths at BASELINE config 1's size (CFG pair) and at the benchmark's full 72 x 128 latent: rel-L2 below
    north_star's 1e-3 for the UNet output and every ControlNet residual; rel-L2 AND element bound below 1.5e-3."""
    cfg, pair = full_pair_f16
    e = _model_errs(cfg, pair, B, 2, h, w)
    for k in ("unet", "controlnet_mid", "controlnet_down"):
        assert e["fp16x2"]["l2"][k] < NORTH_STAR_TOL, e
        assert e["fp16x2"]["both"][k] < 1.5e-3, e


@torch.no_grad()
def test_fullwidth_split_trunk_executors_are_bit_identical(full_pair_f16):
    """Production widths, split trunk: the per-op Python executor (models/blocks.py, lo planes carried by Workspace.trunk)
    and the C++ plan agree bit for bit -- this size runs the C = 320 fused feed-forward and the fused temporal block with
    split residual / output planes, which the tiny configuration does not reach."""
    cfg, pair = full_pair_f16
    _, _, hu, hc = pair
    inputs = make_inputs(cfg, 2, 2, 32, 32, dtype=EL)
    _set_trunk((hu, hc), "fp16x2")
    res = {}
    try:
        for ex in ("plan", "python"):
            hu.executor = hc.executor = ex
            res[ex] = hip_forward(hu, hc, inputs, DEV, with_unet_no_ctrl=False)
    finally:
        hu.executor = hc.executor = "plan"
        _set_trunk((hu, hc), "same")
    for k in ("unet", "mid"):
        assert torch.equal(res["plan"][k], res["python"][k]), k
    assert all(torch.equal(a, b) for a, b in zip(res["plan"]["down"], res["python"]["down"]))
