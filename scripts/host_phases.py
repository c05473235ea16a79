"""Host-side time of the phases of a train step (no device sync inside the loop): where a host-bound small-batch
configuration spends its Python time (development aid)."""
import time, numpy as np, torch, sys
sys.argv=["x"]
from shadow_gnn_amd.minibatch import TRAIN, MinibatchShallowExtractor
from shadow_gnn_amd.models import DeepGNN
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
import torch.nn.functional as F
dev=torch.device("cuda:0")
N,nnz,F0,C=SHAPES["arxiv"]
indptr,indices=make_graph_torch(N,nnz,seed=0,device=dev,max_degree=MAX_DEGREE["arxiv"])
g=torch.Generator(device=dev); g.manual_seed(1)
feat=torch.randn(N,F0,generator=g,device=dev); label=torch.randint(0,C,(N,),generator=g,device=dev)
B=256; steps=200
roots=np.resize(np.random.default_rng(2).permutation(N), B*(steps+5)).astype(np.int64)
mb=MinibatchShallowExtractor.on_device({TRAIN:(indptr,indices)},{TRAIN:roots},dict(method="khop",depth=2,budget=20,add_self_edge=False),("hops",),feat,label,batch_size=B,device=dev,seed_cpp=3)
mb.epoch_start_reset(0,TRAIN); mb.shuffle_entity(TRAIN,perm=np.arange(roots.size))
arch=dict(num_layers=5,num_cls_layers=1,heads=1,dim=256,act="elu",layer_norm="norm_feat",feature_augment_ops="sum",aggr="sage",residue="none",pooling="center",loss="softmax")
m=DeepGNN(F0,F0,C,0,arch,[("hops",mb.get_aug_dim("hops"))],1,dict(dropout=0.25,dropedge=0.15,lr=2e-5),"node").to(dev)
T=dict(batch=0,fwd=0,loss=0,bwd=0,clip=0,opt=0)
for it in range(steps+5):
    t0=time.perf_counter(); bt=mb.one_batch(TRAIN); t1=time.perf_counter()
    m.train(); m.optimizer.zero_grad(set_to_none=True)
    args=bt.to_dict({"feat_ens","adj_ens","target_ens","size_subg_ens","feat_aug_ens"}); args["feat_ens"]=list(args["feat_ens"])
    preds,_=m(TRAIN,dropedge=0.15,**args); t2=time.perf_counter()
    loss=m._loss(preds,F.one_hot(bt.label.long(),C)); t3=time.perf_counter()
    loss.backward(); t4=time.perf_counter()
    torch.nn.utils.clip_grad_norm_(m.parameters(),5); t5=time.perf_counter()
    m.optimizer.step(); t6=time.perf_counter()
    if it>=5:
        for k,v in zip(T,(t1-t0,t2-t1,t3-t2,t4-t3,t5-t4,t6-t5)): T[k]+=v
torch.cuda.synchronize()
print({k:round(v/steps*1e3,3) for k,v in T.items()}, "total", round(sum(T.values())/steps*1e3,3), "ms/step host")
