"""Fuzz of sl_segment_pool_fwd/bwd (mean / max / sum over subgraph rows) against an fp64 torch reference: segment sizes
1 .. 20 000, widths 1 .. 513, ties under max (development aid)."""
import numpy as np, torch
from shadow_gnn_amd import ops
DEV="cuda:0"
rng=np.random.default_rng(3)
bad=0
for trial in range(120):
    P=int(rng.choice([1,2,17,300])); F=int(rng.choice([1,3,4,47,100,256,300,513]))
    sizes=rng.choice([1,1,2,5,63,64,65,1000,20000], P) if P<50 else rng.integers(1,60,P)
    if sizes.sum()>400000: sizes=np.minimum(sizes,2000)
    n=int(sizes.sum()); off=np.concatenate([[0],np.cumsum(sizes)]).astype(np.int32)
    mode=str(rng.choice(["mean","max","sum"]))
    X=torch.randn(n,F,device=DEV); 
    if rng.random()<0.3: X=torch.round(X*2)/2      # ties for max
    x=X.clone().requires_grad_(True); G=torch.randn(P,F,device=DEV)
    out=ops.segment_pool(x, torch.from_numpy(off).to(DEV), mode); (out*G).sum().backward()
    xr=X.clone().double().requires_grad_(True)
    segs=[xr[off[i]:off[i+1]] for i in range(P)]
    ref=torch.stack([{"mean":s.mean(0),"sum":s.sum(0),"max":s.max(0).values}[mode] for s in segs])
    (ref*G.double()).sum().backward()
    e1=float((out.double()-ref).abs().max()/(ref.abs().max()+1e-9)); 
    if mode=="max":
        # ties: any argmax is a valid subgradient; compare the gradient mass per segment/column instead
        gs=torch.stack([x.grad[off[i]:off[i+1]].sum(0) for i in range(P)]); gr=torch.stack([xr.grad[off[i]:off[i+1]].sum(0) for i in range(P)])
        e2=float((gs.double()-gr).abs().max()/(gr.abs().max()+1e-9))
    else:
        e2=float((x.grad.double()-xr.grad).abs().max()/(xr.grad.abs().max()+1e-9))
    if e1>1e-5 or e2>1e-5:
        bad+=1; print("BAD",trial,P,F,mode,sizes[:5],e1,e2)
print("done, bad",bad)
