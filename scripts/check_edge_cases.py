import numpy as np, torch, scipy.sparse as sp
from shadow_gnn_amd import ops, layers
from shadow_gnn_amd.models import DeepGNN
from shadow_gnn_amd.minibatch import OneBatchSubgraph, TRAIN, VALID
DEV="cuda:0"
def csr_of(A):
    return ops.DeviceCSR(torch.from_numpy(A.indptr.astype(np.int32)).to(DEV), torch.from_numpy(A.indices.astype(np.int32)).to(DEV))
def run(name, fn):
    try:
        r = fn(); torch.cuda.synchronize(); print(name, "ok", r if r is not None else "")
    except Exception as ex:
        print(name, "EXC", type(ex).__name__, str(ex)[:160])
# 1. single-node subgraphs without edges, B=1 and B=5
for aggr in ("sage","gcn","gat"):
    for B, n_per, selfloop in ((1,1,False),(5,1,True),(3,2,False)):
        def f():
            blocks=[sp.csr_matrix(np.eye(n_per,dtype=np.float32) if selfloop else np.zeros((n_per,n_per),dtype=np.float32)) for _ in range(B)]
            A=sp.block_diag(blocks,format="csr"); n=A.shape[0]
            arch=dict(num_layers=2,num_cls_layers=1,heads=2 if aggr=="gat" else 1,branch_sharing=False,dim=16,act="relu",layer_norm="norm_feat",feature_augment_ops="sum",aggr=aggr,residue="none",pooling="center",loss="softmax",ensemble_act="relu")
            torch.manual_seed(0)
            m=DeepGNN(8,8,3,0,arch,[],1,dict(lr=0.01,dropout=0.1,dropedge=0.1),"node").to(DEV)
            bt=lambda: OneBatchSubgraph([csr_of(A)],[torch.randn(n,8,device=DEV)],torch.randint(0,3,(B,),device=DEV),torch.full((1,B),n_per,dtype=torch.int64,device=DEV),[(torch.arange(B)*n_per).to(DEV)],[{}])
            l=float(m.step(TRAIN,"running",bt())["loss"].detach()); m.prune_tail=True; l2=float(m.step(TRAIN,"running",bt())["loss"].detach())
            assert np.isfinite(l) and np.isfinite(l2)
            return (round(l,3), round(l2,3))
        run(f"{aggr} B={B} n_per={n_per} selfloop={selfloop}", f)
# 2. raw ops with zero rows
def z():
    X=torch.zeros(0,64,device=DEV); c=ops.DeviceCSR(torch.zeros(1,dtype=torch.int32,device=DEV), torch.zeros(0,dtype=torch.int32,device=DEV))
    y=ops.spmm(ops.adj_norm_rw(c), X); assert y.shape==(0,64)
    o=ops.act_norm([X],["relu"],torch.ones(1,64,device=DEV),torch.zeros(1,64,device=DEV)); assert o.shape==(0,64)
    g=ops.gather_rows(torch.randn(10,64,device=DEV), torch.zeros(0,dtype=torch.int32,device=DEV)); assert g.shape==(0,64)
    return "zero-row ops"
run("empty", z)
