import numpy as np, torch, scipy.sparse as sp
from oracle import layers_oracle as lo
from shadow_gnn_amd import ops
from shadow_gnn_amd.models import DeepGNN
from shadow_gnn_amd.minibatch import OneBatchSubgraph, TRAIN, VALID
DEV="cuda:0"
rng=np.random.default_rng(0)
B, n_per = 40, 300          # 12000 nodes: large enough for the split GEMM (M >= 8192)
blocks=[]
for b in range(B):
    a=(rng.random((n_per,n_per))<0.02).astype(np.float32); a=np.maximum(a,a.T); np.fill_diagonal(a,1.0); blocks.append(sp.csr_matrix(a))
A=sp.block_diag(blocks,format="csr"); A.sort_indices(); n=A.shape[0]
off=np.arange(B+1)*n_per
eoff=A.indptr[off]
for aggr in ("sage","gcn","gat"):
  for dim in (130, 200, 512, 36, 800):
    F0, C = 50, 7
    heads = (4 if dim % 4 == 0 else 2) if aggr=="gat" else 1
    if aggr=="gat" and dim % heads: continue
    arch=dict(num_layers=2,num_cls_layers=1,heads=heads,branch_sharing=False,dim=dim,act="relu",layer_norm="norm_feat",feature_augment_ops="sum",aggr=aggr,residue="none",pooling="center",loss="softmax",ensemble_act="relu")
    torch.manual_seed(1)
    try:
        m=DeepGNN(F0,F0,C,0,arch,[],1,dict(lr=0.01,dropout=0.0,dropedge=0.0),"node").to(DEV)
    except Exception as ex:
        print(aggr, dim, "ctor:", type(ex).__name__, ex); continue
    X=torch.randn(n,F0)
    csr=ops.DeviceCSR(torch.from_numpy(A.indptr.astype(np.int32)).to(DEV),torch.from_numpy(A.indices.astype(np.int32)).to(DEV),
                      subg_off=torch.from_numpy(off.astype(np.int32)).to(DEV),subg_edge_off=torch.from_numpy(eoff.astype(np.int32)).to(DEV),max_subg_nodes=n_per)
    tgt=torch.arange(B)*n_per
    bt=OneBatchSubgraph([csr],[X.to(DEV)],torch.randint(0,C,(B,),device=DEV),torch.full((1,B),n_per,dtype=torch.int64,device=DEV),[tgt.to(DEV)],[{}])
    try:
        out=m.step(VALID,"running",bt)
        p={k:v.detach().cpu() for k,v in m.state_dict().items()}
        ref,_=lo.model_forward(p,arch,X,A.indptr,A.indices,[n_per]*B,tgt.numpy())
        got=out["preds"].cpu()
        refp=torch.softmax(ref,1)
        err=float((got-refp).abs().max())
        bt2=OneBatchSubgraph([csr],[X.to(DEV)],bt.label,bt.size_subg_ens,bt.target_ens,[{}])
        m.step(TRAIN,"running",bt2)
        print(aggr, dim, "ok max|dpreds| %.2e"%err)
    except Exception as ex:
        print(aggr, dim, "FAIL:", type(ex).__name__, str(ex)[:200])
