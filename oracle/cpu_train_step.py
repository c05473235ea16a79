"""CPU PyTorch restatement of the reference's TRAINING STEP for the benchmark architectures (GCN / GraphSAGE
stack, residue 'none', centre pooling, node task) at full batch size.

TEST INFRASTRUCTURE ONLY -- never imported by the product path; bench.py's ``cpu_baseline`` leg times it on
the GPU box's host cores ("the reference's ... CPU PyTorch path", BASELINE.json north_star).  It is a timing
stand-in, not a checker; its forward pass is pinned against oracle/layers_oracle.py (dense adjacency, itself
pinned on the reference's golden vectors) in tests/test_layers_oracle_golden.py -- this file only swaps the
dense matrix for torch.sparse so that a 300 k-node batch fits.

What it follows (line numbers: /root/reference):
  adjacency   D^-1 A / D^-1/2 A D^-1/2 as a torch sparse COO tensor, drop-edge on the values
              (shaDow/frontend/graph_utils.py:67-95, :109-145)
  layer       dropout -> torch.sparse.mm -> nn.Linear + act + norm_feat (+ self branch)   (shaDow/layers.py:326-338,
              :417-444 GCN, :447-494 GraphSAGE)
  read-out    row of the root from the last layer, L2 normalise, MLP classifier with norm_feat
              (shaDow/layers.py:159-163, shaDow/models.py:176-201)
  step        cross-entropy, backward, clip_grad_norm_(5), Adam               (shaDow/models.py:213-246)
"""
import time

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

_ACT = {"relu": F.relu, "elu": F.elu, "tanh": torch.tanh, "I": lambda x: x}


def _norm_feat(x, scale, offset):
    mean = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, unbiased=False, keepdim=True) + 1e-9
    return (x - mean) * scale * torch.rsqrt(var) + offset


class _Layer(nn.Module):
    def __init__(self, kind, dim_in, dim_out, act, dropout):
        super().__init__()
        self.kind, self.act, self.p = kind, _ACT[act], dropout
        nb = 2 if kind == "sage" else 1
        self.lins = nn.ModuleList(nn.Linear(dim_in, dim_out) for _ in range(nb))
        self.scale = nn.Parameter(torch.ones(nb, dim_out))
        self.offset = nn.Parameter(torch.zeros(nb, dim_out))

    def forward(self, x, adj):
        x = F.dropout(x, self.p, self.training)
        agg = torch.sparse.mm(adj, x)
        if self.kind == "gcn":
            return _norm_feat(self.act(self.lins[0](agg)), self.scale[0], self.offset[0])
        return (_norm_feat(self.act(self.lins[0](x)), self.scale[0], self.offset[0])
                + _norm_feat(self.act(self.lins[1](agg)), self.scale[1], self.offset[1]))


class CpuModel(nn.Module):
    def __init__(self, kind, num_layers, dim_in, dim, num_classes, act, dropout, aug_dim=0):
        super().__init__()
        self.layers = nn.ModuleList(_Layer(kind, dim_in if i == 0 else dim, dim, act, dropout) for i in range(num_layers))
        self.cls = nn.Linear(dim, num_classes)
        self.cls_scale = nn.Parameter(torch.ones(num_classes))
        self.cls_offset = nn.Parameter(torch.zeros(num_classes))
        # 'sum' feature augmentation: Linear(one-hot hop encoding) added into the raw features (models.py:183-189)
        self.aug = nn.Linear(aug_dim, dim_in) if aug_dim else None

    def forward(self, x, adj, target, enc=None):
        if self.aug is not None:
            x = x + self.aug(enc)
        for l in self.layers:
            x = l(x, adj)
        emb = F.normalize(x[target], p=2, dim=1)
        return _norm_feat(self.cls(emb), self.cls_scale, self.cls_offset)


def norm_adj(indptr, indices, kind, dropedge, gen):
    """torch sparse COO of the normalised batch adjacency (all-ones data, drop-edge = int(nnz p) positions drawn
    with replacement zeroed, graph_utils.py:85-88)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    n, e = indptr.size - 1, indices.size
    rows = torch.from_numpy(np.repeat(np.arange(n), np.diff(indptr)))
    cols = torch.from_numpy(indices)
    w = torch.ones(e)
    k = int(e * dropedge)
    if k > 0:
        w[torch.randint(0, e, (k,), generator=gen)] = 0
    deg = torch.zeros(n).index_add_(0, rows, w).clamp_(min=1e-12)
    if kind == "gcn":
        d = deg.rsqrt()
        w = w * d[rows] * d[cols]
    else:
        w = w / deg[rows]
    return torch.sparse_coo_tensor(torch.stack([rows, cols]), w, (n, n)).coalesce()


def time_train_steps(indptr, indices, feat, target, label, kind, num_layers, dim, num_classes, act, dropout, dropedge,
                     lr, threads, budget_s=25.0, max_steps=3, enc=None):
    """Runs whole training steps on the CPU until ``budget_s`` is spent (at least one, at most ``max_steps``
    after one untimed warm-up if the budget allows).  Returns (steps, seconds, warmup_seconds)."""
    torch.set_num_threads(int(threads))
    gen = torch.Generator().manual_seed(0)
    torch.manual_seed(0)
    model = CpuModel(kind, num_layers, feat.shape[1], dim, num_classes, act, dropout,
                     aug_dim=(enc.shape[1] if enc is not None else 0))
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    feat, target, label = feat.float(), target.long(), label.long()

    def step():
        model.train()
        adj = norm_adj(indptr, indices, kind, dropedge, gen)
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(model(feat, adj, target, enc), label)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
        opt.step()
        return float(loss.detach())

    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    if warm > budget_s * 0.6:          # one step is all the budget allows: report it (includes first-touch costs)
        return 1, warm, 0.0
    steps, t = 0, 0.0
    while steps < max_steps and t + warm < budget_s:
        t0 = time.perf_counter()
        step()
        t += time.perf_counter() - t0
        steps += 1
    return steps, t, warm
