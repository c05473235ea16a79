import sys; sys.path.insert(0,'/root/repo')
import torch
from shadow_gnn_amd import ops
dev='cuda:0'
for n,F in ((1024,47),(1024,48),(1024,64),(4096,47),(1024,100)):
  for act in (0,1):
    Z=torch.randn(n,F,device=dev); sc=torch.ones(1,F,device=dev); of=torch.zeros(1,F,device=dev); b=torch.zeros(F,device=dev)
    f=lambda: ops._an_fwd([Z],[b],[act],sc,of,F,1.0,(0.0,0))
    f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(n,F,act, f"{e0.elapsed_time(e1)/50*1e3:.1f} us")
