#!/usr/bin/env python
"""Debug: where does ctrlv_ff_fused differ from the two-launch path (per 32-row group x 32-column block)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctrlv_amd import ops, packing
DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
M, C, I = int(os.environ.get("FF_M", 256)), 320, 1280
r = lambda *s: torch.randn(*s, generator=g, device=DEV)
w1p, b1p = packing.pack_geglu(r(2 * I, C) / C ** 0.5, r(2 * I))
w2p = packing.pack_linear(r(C, I) / I ** 0.5)
b1, b2 = b1p.float().contiguous(), r(C)
w1f, w2f = ops.ff_fused_pack(w1p, b1, w2p)
x, r1 = r(M, C).bfloat16(), r(M, C).bfloat16()
u = torch.empty(M, I, dtype=torch.bfloat16, device=DEV)
ref = torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
out = torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
ops.gemm(x, w1p, u, N=2 * I, cin=C, bias=b1, geglu=1)
ops.gemm(u, w2p, ref, N=C, cin=I, bias=b2, R1=r1)
ops.ff_fused(x, w1f, w2f, out, bias=b2, R1=r1)
torch.cuda.synchronize()
d = (out.float() - ref.float()).abs()
print("max abs diff", d.max().item(), "ref rms", ref.float().pow(2).mean().sqrt().item())
blk = d.view(M // 32, 32, C // 32, 32).amax(dim=(1, 3))
torch.set_printoptions(precision=3, linewidth=200)
print(blk[:16])
# h-only probe: zero W2 except chunk c -> which chunks contribute wrongly
for c in (0, 1, 2, 3, 4, 5, 78, 79):
    w2 = torch.zeros(C, I, device=DEV)
    w2[:, 16 * c:16 * c + 16] = r(C, 16)
    w2p_c = packing.pack_linear(w2)
    w1f_c, w2f_c = ops.ff_fused_pack(w1p, b1, w2p_c)
    ops.gemm(u, w2p_c, ref, N=C, cin=I, bias=b2, R1=r1)
    ops.ff_fused(x, w1f_c, w2f_c, out, bias=b2, R1=r1)
    torch.cuda.synchronize()
    dd = (out.float() - ref.float()).abs().view(M // 32, 32, C // 32, 32).amax(dim=(1, 3))
    print("chunk", c, "max diff per (row group, col block):", dd[:4].flatten().tolist()[:40])
