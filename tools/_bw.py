import torch, time
dev="cuda:0"
x=torch.randn(460800,320,device=dev).bfloat16()
o=torch.empty(460800,960,device=dev,dtype=torch.bfloat16)
o2=torch.empty(460800,960,device=dev,dtype=torch.bfloat16)
def t(f,nb,name,reps=20):
    f(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/reps
    print(f"{name:30s} {ms*1e3:8.1f} us  {nb/ms/1e9:6.2f} TB/s")
t(lambda:o.fill_(1.0), o.numel()*2, "fill 885MB (pure write)")
t(lambda:o.zero_(), o.numel()*2, "zero_ 885MB (memset)")
t(lambda:o2.copy_(o), o.numel()*4, "copy 885MB (1R1W)")
t(lambda:torch.cat([x,x,x],1,out=o), o.numel()*2+x.numel()*2, "cat x3 (1R:3W)")
t(lambda:torch.add(o,o2,out=o2), o.numel()*6, "add (2R1W)")
t(lambda:o.sum(), o.numel()*2, "sum (pure read)")
