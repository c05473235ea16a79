"""Edge-list (sparse) restatement of the shaDow model forward for batches too large for the dense
oracle, in a chosen dtype (fp64 by default: the comparison then measures the HIP path's error, not the
checker's).

TEST INFRASTRUCTURE ONLY -- never imported by the product path.  It is pinned twice in the CPU suite
(tests/test_layers_oracle_golden.py): against the reference's golden model steps (tests/golden/
models_step.npz, produced by shaDow/models.py itself) and against the dense oracle/layers_oracle.py on
random batches.  The layer arithmetic is layers_oracle's with the n x n matrix replaced by per-edge
index_add / scatter_reduce; read-out and classifier are shared code (layers_oracle.readout_and_classify).
Line numbers refer to /root/reference.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import layers_oracle as lo


class EdgeList:
    """rows / cols of the batch CSR (duplicates kept: an un-coalesced COO adds them up, as torch.sparse does)."""

    def __init__(self, indptr, indices, dtype=torch.float64):
        indptr = np.asarray(indptr, dtype=np.int64)
        self.n = indptr.size - 1
        self.rows = torch.from_numpy(np.repeat(np.arange(self.n), np.diff(indptr)))
        self.cols = torch.from_numpy(np.asarray(indices, dtype=np.int64).copy())
        self.dtype = dtype
        self.w = torch.ones(self.cols.numel(), dtype=dtype)

    def degree(self):
        return torch.zeros(self.n, dtype=self.dtype).index_add_(0, self.rows, self.w)

    def normalised(self, kind):
        """adj_norm_rw: D^-1 A with D = clamp(row sum, 1) (frontend/graph_utils.py:84-94);
        adj_norm_sym: D^-1/2 A D^-1/2 with D = clip(row sum, 1) (:140-142); GAT keeps the 0/1 values (layers.py:591)."""
        out = EdgeList.__new__(EdgeList)
        out.n, out.rows, out.cols, out.dtype = self.n, self.rows, self.cols, self.dtype
        d = torch.clamp(self.degree(), min=1)
        if kind == "sage":
            out.w = self.w / d[self.rows]
        elif kind == "gcn":
            s = d.pow(-0.5)
            out.w = self.w * s[self.rows] * s[self.cols]
        else:
            out.w = self.w
        return out

    def matmul(self, X):
        """A @ X (torch.sparse.mm, layers.py:326-327,433)"""
        return torch.zeros(self.n, X.shape[1], dtype=X.dtype).index_add_(0, self.rows, X[self.cols] * self.w[:, None])


def _act(act, p, z, keep, stats):
    """The layer's activation.  ``keep`` (optional, relu only): the 0/1 pattern z > 0 of ANOTHER evaluation of the same
    network (the fp32 run under test).  relu has no derivative at 0; among ~10^7 pre-activations of a batch a handful lie
    within rounding of 0 and two precisions put them on different sides -- each such unit switches one weight row's
    gradient by that node's whole contribution, in any implementation.  With ``keep`` the oracle takes the other run's
    side for those units (h = z * keep) so that both differentiate the SAME piecewise-linear function.  ``stats`` reports
    how many units that concerned and the largest |z| among them (the caller bounds both: a side that differs far from 0
    would be a real error, not a kink; rows whose activations are almost all dead have a tiny variance and amplify
    rounding differences of the layer before, so the band is not 1 ulp wide)."""
    if keep is None:
        return lo.act_fn(act, p)(z)
    assert act == "relu", act
    keep = keep.to(torch.bool)
    differ = (z.detach() > 0) != keep
    if stats is not None:
        stats["kink_units"] = stats.get("kink_units", 0) + int(differ.sum())
        stats["units"] = stats.get("units", 0) + differ.numel()
        if bool(differ.any()):
            stats["kink_max_abs_z"] = max(stats.get("kink_max_abs_z", 0.0), float(z.detach().abs()[differ].max()))
    return z * keep.to(z.dtype)


def gcn_forward(p, X, A, act, keep=None, stats=None):
    z = F.linear(A.matmul(X), p["f_lin.weight"], p["f_lin.bias"])                       # layers.py:433-435
    return lo.f_norm(_act(act, p, z, keep[0] if keep else None, stats), p["scale"][0], p["offset"][0])


def sage_forward(p, X, A, act, keep=None, stats=None):
    zs = F.linear(X, p["f_lin_self.weight"], p["f_lin_self.bias"])
    zn = F.linear(A.matmul(X), p["f_lin_neigh.weight"], p["f_lin_neigh.bias"])
    if stats is not None and "z_taps" in stats:           # (diagnostics: the pre-activations incl. bias, per layer)
        stats["z_taps"].append((zs.detach(), zn.detach()))
    hs = _act(act, p, zs, keep[0] if keep else None, stats)   # layers.py:473-483
    hn = _act(act, p, zn, keep[1] if keep else None, stats)
    return lo.f_norm(hs, p["scale"][0], p["offset"][0]) + lo.f_norm(hn, p["scale"][1], p["offset"][1])


def gat_forward(p, X, A, act, heads):
    """GAT.forward + _aggregate_attention (layers.py:560-626) on the edge list."""
    n = X.shape[0]
    hs = lo.act_fn(act, p)(F.linear(X, p["f_lin.0.weight"], p["f_lin.0.bias"])).view(n, heads, -1)
    hn = lo.act_fn(act, p)(F.linear(X, p["f_lin.1.weight"], p["f_lin.1.bias"])).view(n, heads, -1)
    att = p["attention"]
    rows, cols = A.rows, A.cols
    outs_n, outs_s = [], []
    for k in range(heads):
        a_s = F.leaky_relu(hs[:, k] @ att[0, k], 0.2)                                   # :568
        a_n = F.leaky_relu(hn[:, k] @ att[1, k], 0.2)                                   # :569
        e = a_s[rows] + a_n[cols]                                                       # :570
        mx = torch.full((n,), float("-inf"), dtype=e.dtype).scatter_reduce(0, rows, e, "amax", include_self=True)   # :572
        mx = torch.where(torch.isfinite(mx), mx, torch.zeros_like(mx)).detach()
        pexp = torch.exp(e - mx[rows]) * A.w                                            # :574-575
        denom = torch.clamp(torch.zeros(n, dtype=e.dtype).index_add_(0, rows, pexp), min=1e-10)   # :578
        agg = torch.zeros(n, hn.shape[2], dtype=e.dtype).index_add_(0, rows, hn[cols, k] * pexp[:, None]) / denom[:, None]
        outs_n.append(lo.f_norm(agg, p["scale"][0, k], p["offset"][0, k]))              # :620-622 (index 0 = neigh)
        outs_s.append(lo.f_norm(hs[:, k], p["scale"][1, k], p["offset"][1, k]))
    return (torch.cat(outs_s, 1) + torch.cat(outs_n, 1)) / 2                            # :623-625


def model_forward(p, arch, X, indptr, indices, sizes, target, hop1hot=None, dtype=torch.float64, relu_keep=None, stats=None,
                  edge_keep=None, in_drop=None, readout_drop=None):
    """DeepGNN.forward (models.py:169-204, one branch).  ``p``: state_dict tensors (any float dtype; cast to
    ``dtype`` here -- pass leaves of that dtype with requires_grad to get gradients).  ``relu_keep``: per layer, the
    z > 0 patterns of the run under test (one per Linear branch), see _act; ``stats`` collects the kink-unit count.
    Training-mode randomness is taken from the caller (the reference's torch RNG streams cannot be reproduced, the run
    under test hands over ITS draws): ``edge_keep`` -- 0/1 per CSR edge, the drop-edge mask applied before the
    normalisation (graph_utils.py:85-94: dropped positions are zeroed, the degree is the row sum of what is left);
    ``in_drop`` -- per layer, the multiplier tensor of its input nn.Dropout (keep / (1 - p), layers.py:430,471,601) or None;
    ``readout_drop`` -- the same for the read-out MLP's input dropout (layers.py:110; residue / pooling read-outs only)."""
    kind, L, heads, act = arch["aggr"], arch["num_layers"], int(arch.get("heads", 1)), arch["act"]
    p = {k: (v if v.dtype == dtype else v.to(dtype)) for k, v in p.items()}
    x = X.to(dtype)
    if hop1hot is not None:                                                              # models.py:185-189
        x = x + F.linear(hop1hot.to(dtype), p["aug_layers.0.0.weight"], p["aug_layers.0.0.bias"])
    A = EdgeList(indptr, indices, dtype)
    if edge_keep is not None:
        A.w = torch.as_tensor(edge_keep).to(dtype)
    A = A.normalised(kind)
    feats = []
    for l in range(L):
        if in_drop is not None and in_drop[l] is not None:
            x = x * in_drop[l].to(dtype)
        pre = f"conv_layers.0.{l}."
        lp = {k[len(pre):]: v for k, v in p.items() if k.startswith(pre)}
        keep = relu_keep[l] if relu_keep is not None else None
        if kind == "gcn":
            x = gcn_forward(lp, x, A, act, keep, stats)
        elif kind == "sage":
            x = sage_forward(lp, x, A, act, keep, stats)
        else:
            x = gat_forward(lp, x, A, act, heads)
        feats.append(x)
    return lo.readout_and_classify(p, arch, feats, sizes, target, readout_drop=readout_drop)
