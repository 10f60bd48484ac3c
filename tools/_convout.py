import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops, packing
DEV="cuda:0"; g=torch.Generator(device=DEV).manual_seed(0)
r=lambda *s: torch.randn(*s, generator=g, device=DEV)
H,W,C,n=72,128,320,50; M=n*H*W
x=[r(M,C).bfloat16() for _ in range(3)]
w=packing.pack_conv3x3(torch.randn(4,C,3,3)/ (9*C)**0.5).to(DEV)
print('packed', w.shape)
bias=r(w.shape[0])
outs=[torch.empty(M,4,dtype=torch.bfloat16,device=DEV) for _ in range(3)]
ref=None
for tile in (0,1,3,4,10):
    try:
        def run(i): ops.gemm(x[i], w, outs[i], N=w.shape[0], cin=C, taps=9, mode=1, conv=(H,W,H,W,1,0), bias=bias, n_store=4, tile=tile)
        for i in range(3): run(i)
        torch.cuda.synchronize()
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            for i in range(3): run(i)
        e.record(); torch.cuda.synchronize()
        o=outs[0].float().clone()
        if ref is None: ref=o
        print('tile',tile, f"{s.elapsed_time(e)/9*1e3:.1f} us", 'maxdiff vs tile0', float((o-ref).abs().max()))
    except Exception as ex:
        print('tile',tile,'error',str(ex)[:150])
