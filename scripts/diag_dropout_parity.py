"""Diagnostic: the dropout / drop-edge ON parity step (tests/test_layers_gpu.py::test_timed_configuration_...) layer by layer --
max |z_run - z_fp64| per layer and branch, the rows where it is largest and their normalisation statistics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests.test_layers_gpu import _bench_scale_batch, DEV
from oracle import model_oracle_sparse as mos
from shadow_gnn_amd import ops
from shadow_gnn_amd.minibatch import OneBatchSubgraph, TRAIN
from shadow_gnn_amd.models import DeepGNN

act = sys.argv[1] if len(sys.argv) > 1 else "relu"
lazy_on = os.environ.get("DIAG_LAZY", "1") == "1"
P_DROP, P_EDGE, L = float(os.environ.get("DIAG_PDROP", "0.4")), float(os.environ.get("DIAG_PEDGE", "0.05")), 5
b, X, labels, F0, C = _bench_scale_batch("sage", 128)
n = b.num_nodes
arch = dict(num_layers=L, num_cls_layers=1, heads=1, dim=256, act=act, layer_norm="norm_feat", feature_augment_ops="sum",
            aggr="sage", residue="none", pooling="center", loss="softmax")
torch.manual_seed(41)
model = DeepGNN(F0, F0, C, 0, arch, [], 1, dict(dropout=P_DROP, dropedge=P_EDGE, lr=0.002), "node").to(DEV)
with torch.no_grad():
    for q in model.parameters():
        q.add_(0.05 * torch.randn_like(q))
model.optimizer = torch.optim.SGD(model.parameters(), lr=0.0)
p0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
table = X.to(DEV)
feat = ops.LazyRows(table, torch.arange(n, device=DEV, dtype=torch.int32)) if lazy_on else table
adj = ops.DeviceCSR(b.indptr, b.indices, subg_off=b.subg_node_off, subg_edge_off=b.subg_edge_off, max_subg_nodes=b.counts["max_subg_nodes"])
batch = OneBatchSubgraph([adj], [feat], labels.to(DEV), b.size_subg.unsqueeze(0), [b.target], [{}])
seeds, masks = [], []
rs, rm = ops.new_dropout_seed, ops.dropedge_mask
def ls():
    s_ = rs(); seeds.append(s_); return s_
def lm(csr, de, symmetric=False):
    m = rm(csr, de, symmetric); masks.append(m); return m
ops.new_dropout_seed, ops.dropedge_mask = ls, lm
ops.Z_TAP = []
ret = model.step(TRAIN, "running", batch)
torch.cuda.synchronize()
tap = ops.Z_TAP; ops.Z_TAP = None
print("seeds", len(seeds), "masks", len(masks))
widths = [F0] + [256] * (L - 1)
in_drop = [ops.dropout_keep_mask(n, w, P_DROP, s_, DEV).cpu().double() / (1.0 - P_DROP) for w, s_ in zip(widths, seeds)] if P_DROP > 0 else None
ek = masks[0].cpu() if masks and masks[0] is not None else None
h = b.to_host(); sizes = np.diff(h["subg_node_off"].astype(np.int64))
relu_keep = [[((z + (bb if bb is not None else 0)) > 0).cpu() for z, bb in zip(zs, bs_)] for zs, bs_ in tap[:L]] if act == "relu" else None
st = {"z_taps": []}
p = {k: v.double() for k, v in p0.items()}
with torch.no_grad():
    preds_ref, emb_ref = mos.model_forward(p, arch, X, h["indptr"], h["indices"], sizes, h["target"], relu_keep=relu_keep, stats=st, edge_keep=ek, in_drop=in_drop)
print({k: v for k, v in st.items() if k != "z_taps"})
for l in range(L):
    for br in range(2):
        z_run = (tap[l][0][br] + (tap[l][1][br] if tap[l][1][br] is not None else 0)).cpu().double()
        z_ref = st["z_taps"][l][br]
        d = (z_run - z_ref).abs()
        rowmax = d.max(dim=1).values
        worst = torch.topk(rowmax, 5)
        print(f"layer {l} branch {br}: max|dz| {float(d.max()):.3e}  rows>1e-4: {int((rowmax > 1e-4).sum())}  worst rows {worst.indices.tolist()} {['%.2e' % v for v in worst.values.tolist()]}")
        if l > 0 and float(d.max()) > 1e-4:
            r = int(worst.indices[0])
            # statistics of the layer below at that row (fp64 oracle): h = act(z) per branch
            for bb in range(2):
                zb = st["z_taps"][l - 1][bb][r]
                hb = torch.relu(zb) if act == "relu" else torch.nn.functional.elu(zb)
                print(f"    row {r} layer {l-1} branch {bb}: alive {int((hb > 0).sum())} max h {float(hb.max()):.3e} var {float(hb.var(unbiased=False)):.3e}")
emb = ret["emb_ens"][0].detach().cpu().double()
print("emb max diff", float((emb - emb_ref).abs().max()))
