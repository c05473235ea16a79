"""CPU restatement of the shaDow layer math (plain torch fp32, dense adjacency).

TEST INFRASTRUCTURE ONLY -- never imported by the product path.  Pinned against
golden vectors produced by the reference's own shaDow/layers.py and
shaDow/models.py (tests/golden/layers_fwd_bwd.npz, models_step.npz; generated
by oracle/gen_golden_layers.py).  Line numbers refer to /root/reference.

Everything is written for clarity on tiny inputs: the block-diagonal batch CSR
is expanded to a dense n x n matrix and autograd provides the gradients.
"""
import numpy as np
import torch
import torch.nn.functional as F

ACT = {
    "I": lambda x: x,                                   # LeakyReLU(negative_slope=1), layers.py:28
    "relu": F.relu, "elu": F.elu, "tanh": torch.tanh,
    "leakyrelu": lambda x: F.leaky_relu(x, 0.2),
}


def act_fn(act, p, key="act.weight"):
    """The layer's activation module (layers.py:26-39): fixed functions, or nn.PReLU with its learnable slope
    ("prelu": one per layer, "prelu+": one per output channel) stored in the layer's state_dict as `key`."""
    if act in ("prelu", "prelu+"):
        return lambda x: F.prelu(x, p[key])
    return ACT[act]


def dense_adj(indptr, indices, edge_w=None):
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    n = indptr.size - 1
    rows = np.repeat(np.arange(n), np.diff(indptr))
    A = torch.zeros(n, n)
    w = torch.ones(indices.size) if edge_w is None else torch.as_tensor(edge_w, dtype=torch.float32)
    A.index_put_((torch.as_tensor(rows), torch.as_tensor(indices)), w, accumulate=True)   # duplicates add up (uncoalesced COO)
    return A


def adj_norm_rw(A):
    """D^-1 A, D = clamp(row sum, 1)  (frontend/graph_utils.py:84-94)"""
    return A / torch.clamp(A.sum(1, keepdim=True), min=1)


def adj_norm_sym(A):
    """D^-1/2 A D^-1/2, D = clip(row sum, 1)  (frontend/graph_utils.py:140-142)"""
    d = torch.clamp(A.sum(1), min=1).pow(-0.5)
    return d[:, None] * A * d[None, :]


def f_norm(x, scale, offset):
    """(x - mean) * scale * rsqrt(var_biased + 1e-9) + offset over the last dim (layers.py:334-336)"""
    mean = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, unbiased=False, keepdim=True) + 1e-9
    return (x - mean) * scale * torch.rsqrt(var) + offset


def gcn_forward(p, X, A_norm, act):
    """GCN.forward, layers.py:433-435: aggregate, Linear, act, norm"""
    h = A_norm @ X
    z = F.linear(h, p["f_lin.weight"], p["f_lin.bias"])
    return f_norm(act_fn(act, p)(z), p["scale"][0], p["offset"][0])


def sage_forward(p, X, A_norm, act):
    """GraphSAGE.forward, layers.py:473-483"""
    hs = act_fn(act, p)(F.linear(X, p["f_lin_self.weight"], p["f_lin_self.bias"]))
    hn = act_fn(act, p)(F.linear(A_norm @ X, p["f_lin_neigh.weight"], p["f_lin_neigh.bias"]))
    return f_norm(hs, p["scale"][0], p["offset"][0]) + f_norm(hn, p["scale"][1], p["offset"][1])


def gat_forward(p, X, A_mask, act, heads):
    """GAT.forward + _aggregate_attention, layers.py:560-626.  A_mask: dense 0/1
    (multiplicities for duplicate edges) adjacency, un-normalised (:591)."""
    n = X.shape[0]
    hs = act_fn(act, p)(F.linear(X, p["f_lin.0.weight"], p["f_lin.0.bias"])).view(n, heads, -1)
    hn = act_fn(act, p)(F.linear(X, p["f_lin.1.weight"], p["f_lin.1.bias"])).view(n, heads, -1)
    att = p["attention"]
    outs_n, outs_s = [], []
    present = A_mask > 0
    for k in range(heads):
        a_s = F.leaky_relu(hs[:, k] @ att[0, k], 0.2)            # :568
        a_n = F.leaky_relu(hn[:, k] @ att[1, k], 0.2)            # :569
        e = a_s[:, None] + a_n[None, :]                          # :570
        e_m = torch.where(present, e, torch.full_like(e, float("-inf")))
        mx = e_m.max(dim=1, keepdim=True).values                 # :572
        mx = torch.where(torch.isfinite(mx), mx, torch.zeros_like(mx))
        pexp = torch.exp(e - mx) * A_mask                        # :574-575
        denom = torch.clamp(pexp.sum(1, keepdim=True), min=1e-10)  # :578
        agg = (pexp @ hn[:, k]) / denom                          # :580-581
        outs_n.append(f_norm(agg, p["scale"][0, k], p["offset"][0, k]))      # :620-622 (index 0 = neigh)
        outs_s.append(f_norm(hs[:, k], p["scale"][1, k], p["offset"][1, k]))
    return (torch.cat(outs_s, 1) + torch.cat(outs_n, 1)) / 2     # :623-625


def layer_forward(kind, p, X, indptr, indices, act, heads=1, first=True):
    A = dense_adj(indptr, indices)
    if kind == "gcn":
        return gcn_forward(p, X, adj_norm_sym(A), act)
    if kind == "sage":
        return sage_forward(p, X, adj_norm_rw(A), act)
    if kind == "gat":
        return gat_forward(p, X, A, act, heads)
    raise ValueError(kind)


def hop2onehot(hop, dim):
    """EntityEncoding.hop2onehot_vec, frontend/graph.py:134-147"""
    hop = np.asarray(hop, dtype=np.int64)
    ret = np.zeros((hop.size, dim), dtype=np.float32)
    for i in [-1, 0] + list(range(1, dim - 1)):
        ret[np.where(hop == i)[0], i + 1] = 1
    ret[np.where(hop >= 255)[0], 0] = 1
    return ret


def model_forward(p, arch, X, indptr, indices, sizes, target, hop1hot=None):
    """DeepGNN.forward, models.py:169-204 (one ensemble branch)."""
    kind, L, heads, act = arch["aggr"], arch["num_layers"], int(arch["heads"]), arch["act"]
    x = X
    if hop1hot is not None:                                      # models.py:185-189
        x = x + F.linear(hop1hot, p["aug_layers.0.0.weight"], p["aug_layers.0.0.bias"])
    A = dense_adj(indptr, indices)
    A_n = {"gcn": adj_norm_sym, "sage": adj_norm_rw, "gat": lambda a: a}[kind](A)
    feats = []
    for l in range(L):
        lp = {k[len(f"conv_layers.0.{l}."):]: v for k, v in p.items() if k.startswith(f"conv_layers.0.{l}.")}
        if kind == "gcn":
            x = gcn_forward(lp, x, A_n, act)
        elif kind == "sage":
            x = sage_forward(lp, x, A_n, act)
        else:
            x = gat_forward(lp, x, A_n, act, heads)
        feats.append(x)
    return readout_and_classify(p, arch, feats, sizes, target)


def readout_and_classify(p, arch, feats, sizes, target, readout_drop=None):
    """ResPool + L2 normalisation + classifier on the per-layer outputs ``feats`` (layers.py:154-199,
    models.py:198-204); shared by the dense model above and oracle/model_oracle_sparse.py.  ``readout_drop``: the
    multiplier tensor (keep / (1 - p)) of the read-out's own nn.Dropout in training mode (layers.py:110), taken from the
    run under test -- the reference's torch RNG stream cannot be reproduced."""
    act = arch["act"]
    tgt = torch.as_tensor(np.asarray(target, dtype=np.int64))
    type_res, type_pool = arch["residue"], arch["pooling"]
    if type_pool == "center" and type_res == "none":             # layers.py:159-163
        emb = feats[-1][tgt]
    else:
        sizes_t = torch.as_tensor(np.asarray(sizes, dtype=np.int64))
        off = torch.cumsum(sizes_t, 0) - sizes_t

        def pool(f):                                             # F.embedding_bag, layers.py:175,180
            outs = []
            for s, o in zip(sizes_t.tolist(), off.tolist()):
                seg = f[o:o + s]
                outs.append({"mean": seg.mean(0), "max": seg.max(0).values, "sum": seg.sum(0)}[type_pool])
            return torch.stack(outs)

        def residue(fl):                                         # layers.py:120-130
            if type_res in ("cat", "concat"):
                return torch.cat(fl, 1)
            if type_res == "sum":
                return torch.stack(fl).sum(0)
            return torch.stack(fl).max(0).values
        if type_pool == "center":
            feat_in = residue([f[tgt] for f in feats])
        elif type_res == "none":
            feat_in = torch.cat([feats[-1][tgt], pool(feats[-1])], 1)
        else:
            feat_in = torch.cat([residue([f[tgt] for f in feats]), residue([pool(f) for f in feats])], 1)
        if readout_drop is not None:
            feat_in = feat_in * readout_drop.to(feat_in.dtype)
        z = act_fn(act, p, "res_pool_layers.0.nn.2.weight")(
            F.linear(feat_in, p["res_pool_layers.0.nn.1.weight"], p["res_pool_layers.0.nn.1.bias"]))
        emb = f_norm(z, p["res_pool_layers.0.scale"], p["res_pool_layers.0.offset"])    # layers.py:114-118,199
    emb = F.normalize(emb, p=2, dim=1)                           # models.py:200
    z = F.linear(emb, p["classifier.0.f_lin.weight"], p["classifier.0.f_lin.bias"])
    preds = f_norm(z, p["classifier.0.scale"][0], p["classifier.0.offset"][0])          # MLP act 'I' + norm_feat
    return preds, emb


def model_loss(preds, labels):
    """CrossEntropy on integer labels (models.py:163-166)"""
    return F.cross_entropy(preds, torch.as_tensor(np.asarray(labels, dtype=np.int64)))
