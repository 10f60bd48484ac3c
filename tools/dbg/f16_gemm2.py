import math, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
from ctrlv_amd import ops, packing
DEV = "cuda:0"
g = lambda s: torch.Generator().manual_seed(s)
EL = torch.float16
for (M, N, K) in [(300, 320, 128), (1000, 256, 320), (512, 512, 640)]:
    with packing.element_dtype(EL):
        A = torch.randn(M, K, generator=g(1)).to(EL)
        Wt = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
        Wp = packing.pack_linear(Wt)
        R1 = torch.randn(M, N, generator=g(4)).to(EL)
    lin = A.float() @ Wp[:N].float().T
    for rep in range(3):
        for epi in ("none", "r1"):
            out = torch.full((M, N), 7.0, dtype=EL, device=DEV)
            kw = dict(R1=R1.to(DEV)) if epi == "r1" else {}
            ops.gemm(A.to(DEV), Wp.to(DEV), out, N=N, cin=K, tile=5, **kw)
            torch.cuda.synchronize()
            ref = lin + (R1.float() if epi == "r1" else 0)
            o = out.float().cpu()
            bad = ((o - ref).abs() > 0.02) | o.isnan()
            idx = bad.nonzero().tolist()
            print((M, N, K), epi, "rep", rep, "bad", len(idx), [(r, c, float(o[r, c]), round(float(ref[r, c]), 3)) for r, c in idx[:12]])
