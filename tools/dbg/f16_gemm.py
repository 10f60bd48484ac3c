import math, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
from ctrlv_amd import ops, packing
from tests.parity_utils import rel_l2, max_err
DEV = "cuda:0"
g = lambda s: torch.Generator().manual_seed(s)
for EL in (torch.bfloat16, torch.float16):
    for (M, N, K) in [(300, 320, 128), (1000, 256, 320)]:
        for tile in (1, 5, 6):
            with packing.element_dtype(EL):
                A = torch.randn(M, K, generator=g(1)).to(EL)
                Wt = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
                bias = torch.randn(N, generator=g(3))
                Wp = packing.pack_linear(Wt)
            lin = A.float() @ Wp[:N].float().T
            out = torch.empty(M, N, dtype=EL, device=DEV)
            ops.gemm(A.to(DEV), Wp.to(DEV), out, N=N, cin=K, tile=tile)
            torch.cuda.synchronize()
            d = (out.float().cpu() - lin).abs()
            i = int(d.argmax())
            print(EL, (M, N, K), "tile", tile, "nobias rel", f"{rel_l2(out, lin):.2e}", "max", f"{max_err(out, lin):.2e}",
                  "worst at", divmod(i, N), float(out.float().cpu().flatten()[i]), float(lin.flatten()[i]))
            ops.gemm(A.to(DEV), Wp.to(DEV), out, N=N, cin=K, bias=bias.to(DEV), tile=tile)
            torch.cuda.synchronize()
            print("      bias rel", f"{rel_l2(out, lin + bias):.2e}", "max", f"{max_err(out, lin + bias):.2e}")
