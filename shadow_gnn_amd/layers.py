"""GNN layers with the reference's plugin protocol (shaDow/layers.py), computing
on the MI355X through libshadow_hip.so.

Same class names, constructor arguments, parameter names/shapes (the checkpoint
contract: f_lin, f_lin_self, f_lin_neigh, f_lin.{0,1}, attention, scale, offset)
and the same layer protocol as the reference:

    forward((feat_in, adj, is_normed, dropedge), sizes_subg)
        -> (feat_out, adj_norm, True, 0.)

Layer 0 receives the batch adjacency un-normalised -- a scipy CSR (as in the
reference) or a device-resident ``ops.DeviceCSR`` (fast path) -- normalises it
once, and threads the ``ops.NormAdj`` through the later layers.

Aggregation (SpMM), activation + feature normalisation and the GAT edge softmax
are hand-written HIP kernels; the dense feature x weight products stay on
rocBLAS / hipBLASLt through ``torch.nn.functional.linear``.
"""
from collections import namedtuple

import scipy.sparse as sp
import torch
import torch.nn.functional as F
from torch import nn

from . import ops

Dims_X = namedtuple('Dims_X', ['num_nodes', 'num_feats'])
Dims_adj = namedtuple('Dims_adj', ['num_nodes', 'num_edges'])

# activations of the reference's F_ACT table (shaDow/layers.py:26-34).  The parameter-free ones run inside the
# fused kernels.  'prelu' (nn.PReLU(): one learnable slope per layer) and 'prelu+' (one per output channel) are
# applied by torch between the Linear and the fused normalisation -- the kernels then see the identity -- so the
# slope's gradient comes from autograd and the module keeps the reference's state_dict key ("act.weight").
SUPPORTED_ACT = ("relu", "I", "elu", "tanh", "leakyrelu", "prelu", "prelu+")
LEARNED_ACT = ("prelu", "prelu+")


def _check_act(act):
    if act not in SUPPORTED_ACT:
        raise NotImplementedError(f"activation {act!r} is not provided by the HIP layers {SUPPORTED_ACT}")
    return act


def _learned_act(act, dim_out):
    """nn.PReLU for the learnable activations (get_torch_act, layers.py:37-39), None otherwise."""
    if act == "prelu":
        return nn.PReLU()
    if act == "prelu+":
        return nn.PReLU(num_parameters=dim_out)
    return None


def _as_device_csr(adj, device):
    if isinstance(adj, ops.DeviceCSR):
        return adj
    if isinstance(adj, sp.csr_matrix):
        return ops.DeviceCSR.from_scipy(adj, device)
    raise TypeError(f"layer 0 expects a scipy CSR or a DeviceCSR adjacency, got {type(adj)}")


class EnsembleDummy(nn.Module):
    """Pass-through "ensembler" of a single-branch model (the role of shaDow/layers.py:42-53); the constructor
    accepts and ignores the ensembler arguments so that it can stand in for a real one."""
    def __init__(self, dim_in=0, dim_out=0, **kwargs):
        super().__init__()

    @staticmethod
    def _only(branches):
        if len(branches) != 1:
            raise ValueError(f"EnsembleDummy passes one branch through, got {len(branches)}")
        return branches[0]

    def forward(self, Xi):
        return self._only(Xi)

    def complexity(self, dims):
        return self._only(dims), 0


class shaDowLayer(nn.Module):
    """Parent of the message-passing layers (shaDow/layers.py:299-373)."""
    def __init__(self, dim_in, dim_out, dropout=0.0, act='relu', norm='norm_feat', norm_dim=None, **kwargs):
        """``norm_dim``: shape of the layer's ``scale`` / ``offset`` parameters -- (branches, ..., features of one
        normalisation segment); subclasses pass theirs, the default is one branch over all output features."""
        super().__init__()
        self.dropout = dropout
        self.dim_in, self.dim_out = dim_in, dim_out
        self.act_name = _check_act(act)
        self.act = _learned_act(act, dim_out)        # registered only when it has parameters ("act.weight")
        self.kact = 'I' if self.act is not None else self.act_name      # what the fused kernels apply
        self.f_dropout = nn.Dropout(p=self.dropout)
        # Dropout fusion (set per step by DeepGNN.forward while training): the layer BEFORE this one may
        # already have applied this layer's input dropout inside its act_norm kernel (input_pre_dropped),
        # and this layer may apply the NEXT layer's input dropout to its own output (out_dropout).
        # out_dual: something else reads this layer's un-dropped output too (residue / pooling read-outs), so the
        # kernel writes both; the dropped tensor waits in `dropped_out` for the model loop to hand it on.
        self.input_pre_dropped = False
        self.out_dropout = 0.0
        self.out_dual = False
        self.dropped_out = None
        # set per step by DeepGNN._plan_dropout_fusion: the next layer is the ONLY reader of this layer's output, so the two
        # layers' backward passes may be chained (ops.ChainLink; GraphSAGE only)
        self.chain_next = False
        self.pool_only = False
        # True (set per step by DeepGNN._plan_dropout_fusion): only a row-selecting read-out reads this layer's output
        self.roots_only = False
        # True (set per step by DeepGNN._plan_dropout_fusion): only the next GAT layer's paired Linear reads this output
        self.rows_next = False
        if norm not in ('norm_feat', 'none'):
            raise NotImplementedError("only norm in {'norm_feat', 'none'} (the reference's pairnorm path is unfinished, layers.py:358)")
        self.norm = norm
        self.norm_dim = tuple(norm_dim) if norm_dim is not None else (1, dim_out)
        if norm == 'norm_feat':
            self.offset = nn.Parameter(torch.zeros(self.norm_dim))
            self.scale = nn.Parameter(torch.ones(self.norm_dim))

    def spmm(self, adj, X):
        return ops.spmm(adj, X)

    def in_dropout(self, feat_in):
        """nn.Dropout on the layer input (layers.py:430,471,601) unless the producer fused it."""
        feat_in = ops.dense_rows(feat_in)
        return feat_in if (self.input_pre_dropped and self.training) else self.f_dropout(feat_in)

    def _in_p(self):
        """Probability of the input dropout still to be applied (0 in evaluation / when the producer fused it)."""
        return float(self.dropout) if (self.training and not self.input_pre_dropped) else 0.0

    def can_fuse_out_dropout(self):
        return self.norm == 'norm_feat' and ops.can_fuse_out_dropout(self.dim_out, getattr(self, 'dim_slice', None))

    def _out_p(self):
        return self.out_dropout if self.training else 0.0

    def _drop_kw(self):
        p = self._out_p()
        return {'out_dropout': p, 'dual': bool(self.out_dual and p > 0.0)}

    def _emit(self, res):
        """Fused kernels return out, or (out, dropout(out)) in dual mode: keep the second for the next layer."""
        if isinstance(res, tuple):
            self.dropped_out = res[1]
            return res[0]
        return res

    def take_dropped_out(self):
        d, self.dropped_out = self.dropped_out, None
        return d

    def f_lin_act_norm(self, Xs, lins, acts):
        """sum_b norm_b(act_b(lin_b(X_b))): Linear (rocBLAS) + bias/act/norm/add (one HIP kernel),
        one autograd node with fused bias / scale / offset gradients."""
        if self.act is not None:              # learnable activation: Linear, PReLU (torch), fused norm of the result
            return self.f_act_norm([self.act(ops.linear(x, l)) for x, l in zip(Xs, lins)], ['I'] * len(Xs))
        if self.norm == 'norm_feat':
            return self._emit(ops.linear_act_norm(Xs, lins, acts, self.scale, self.offset, **self._drop_kw()))
        return self.f_act_norm([ops.linear(x, l) for x, l in zip(Xs, lins)], acts)

    def f_act_norm(self, Zs, acts, seg=None, out_scale=1.0):
        """sum_b norm_b(act_b(Z_b)) * out_scale -- the reference's act + f_norm + add
        sequence (layers.py:435, :476-483, :620-625) in one kernel."""
        if self.norm == 'norm_feat':
            return self._emit(ops.act_norm(Zs, acts, self.scale, self.offset, seg=seg, out_scale=out_scale,
                                           **self._drop_kw()))
        out = None
        for z, a in zip(Zs, acts):
            h = _torch_act(a, z)
            out = h if out is None else out + h
        return out * out_scale


def _dense_ops(num_rows, lin):
    """Multiply-accumulates of ``lin`` applied to ``num_rows`` rows (the unit of the reference's complexity())."""
    return int(num_rows) * int(lin.weight.numel())


def _torch_act(act, x):
    if act == 'I':
        return x
    if act == 'relu':
        return F.relu(x)
    if act == 'elu':
        return F.elu(x)
    if act == 'tanh':
        return torch.tanh(x)
    if act == 'leakyrelu':
        return F.leaky_relu(x, 0.2)
    raise NotImplementedError(act)


class MLP(shaDowLayer):
    """shaDow/layers.py:376-400"""
    def __init__(self, dim_in, dim_out, dropout=0.0, act="relu", norm='norm_feat', **kwargs):
        kwargs.pop('norm_dim', None)
        super().__init__(dim_in, dim_out, dropout=dropout, act=act, norm=norm, norm_dim=(1, dim_out), **kwargs)
        self.f_lin = nn.Linear(dim_in, dim_out)

    def forward(self, feat_in):
        feat_in = self.in_dropout(feat_in)
        return self.f_lin_act_norm([feat_in], [self.f_lin], [self.act_name])

    def complexity(self, dims_x):
        """(output dims, multiply-accumulates) -- shaDow/layers.py:396-400."""
        if dims_x.num_feats != self.f_lin.in_features:
            raise ValueError(f"MLP expects {self.f_lin.in_features} input features, got {dims_x.num_feats}")
        return Dims_X(dims_x.num_nodes, self.f_lin.out_features), _dense_ops(dims_x.num_nodes, self.f_lin)


class GCN(shaDowLayer):
    """shaDow/layers.py:417-444 -- aggregate first, then Linear."""
    def __init__(self, dim_in, dim_out, dropout=0.0, act="relu", norm='norm_feat', **kwargs):
        kwargs.pop('norm_dim', None)
        super().__init__(dim_in, dim_out, dropout=dropout, act=act, norm=norm, norm_dim=(1, dim_out), **kwargs)
        self.f_lin = nn.Linear(dim_in, dim_out, bias=True)

    def norm_adj(self, adj, is_normed, dropedge, device):
        if not is_normed and adj is not None:
            # self-edges are already added by the sampler (shaDow/utils.py:126-131)
            return ops.adj_norm_sym(_as_device_csr(adj, device), dropedge=dropedge)
        assert adj is None or isinstance(adj, ops.NormAdj)
        return adj

    def forward(self, inputs, sizes_subg):
        feat_in, adj, is_normed, dropedge = inputs
        adj_norm = self.norm_adj(adj, is_normed, dropedge, feat_in.device)
        if isinstance(feat_in, ops.LazyRows) and ops.FUSE_GATHER_INTO_SPMM and ops.can_fuse_gather(adj_norm, feat_in):
            # layer 0 of the fast path: feature gather + input dropout inside the aggregation kernel
            feat_aggr, _x, _seed = ops.spmm_gather(adj_norm, feat_in, drop_p=self._in_p(), want_dense=False)
        else:
            # gather + input dropout in one pass into line-padded rows (the measured faster form), or the previous layer's output
            x = feat_in.gather_dropped(self._in_p())[0] if isinstance(feat_in, ops.LazyRows) else self.in_dropout(feat_in)
            if (self.act is None and self.norm == 'norm_feat' and self.act_name in ops.ACT_CODE
                    and ops._GcnDense.fusable(x, adj_norm, self.f_lin.weight)):
                # the whole layer as one autograd node, one C call per direction (small batches are host-bound)
                feat_out = self._emit(ops.gcn_dense(x, adj_norm, self.f_lin, self.act_name, self.scale, self.offset, **self._drop_kw()))
                return feat_out, adj_norm, True, 0.
            feat_aggr = self.spmm(adj_norm, x)
        feat_out = self.f_lin_act_norm([feat_aggr], [self.f_lin], [self.act_name])
        return feat_out, adj_norm, True, 0.

    def forward_rows(self, feat_in, adj_norm, level):
        """The layer on the output rows of ``level`` only (tail.py): same arithmetic per row."""
        from . import tail
        feat_in = self.in_dropout(feat_in)
        _, feat_aggr = tail.rect_gather_spmm(feat_in, level, adj_norm)
        return self.f_lin_act_norm([feat_aggr], [self.f_lin], [self.act_name])

    def complexity(self, dims_x, dims_adj):
        """((feature dims, adjacency dims) after the layer, multiply-accumulates): one aggregation at the input
        width plus one Linear (shaDow/layers.py:438-444)."""
        macs = dims_adj.num_edges * dims_x.num_feats + _dense_ops(dims_x.num_nodes, self.f_lin)
        return (Dims_X(dims_x.num_nodes, self.f_lin.out_features), Dims_adj(*dims_adj)), macs


class GraphSAGE(shaDowLayer):
    """shaDow/layers.py:447-494"""
    def __init__(self, dim_in, dim_out, dropout=0.0, act="relu", norm='norm_feat', **kwargs):
        kwargs.pop('norm_dim', None)
        # two normalised branches: row 0 of scale / offset belongs to the self branch, row 1 to the neighbours
        super().__init__(dim_in, dim_out, dropout=dropout, act=act, norm=norm, norm_dim=(2, dim_out), **kwargs)
        for name in ("f_lin_self", "f_lin_neigh"):
            setattr(self, name, nn.Linear(dim_in, dim_out))

    def norm_adj(self, adj, is_normed, dropedge, device):
        if not is_normed and adj is not None:
            return ops.adj_norm_rw(_as_device_csr(adj, device), dropedge=dropedge)
        assert adj is None or isinstance(adj, ops.NormAdj)
        return adj

    def forward_rows(self, feat_in, adj_norm, level):
        """The layer on the output rows of ``level`` only (tail.py): same arithmetic per row."""
        from . import tail
        feat_in = self.in_dropout(feat_in)
        feat_self, feat_neigh = tail.rect_gather_spmm(feat_in, level, adj_norm)
        return self.f_lin_act_norm([feat_self, feat_neigh], [self.f_lin_self, self.f_lin_neigh],
                                   [self.act_name, self.act_name])

    def forward(self, inputs, sizes_subg):
        feat_in, adj, is_normed, dropedge = inputs
        adj_norm = self.norm_adj(adj, is_normed, dropedge, feat_in.device)
        fused = self.norm == 'norm_feat' and self.f_lin_self.weight.shape[0] % 4 == 0 and self.act is None
        if fused and isinstance(feat_in, ops.LazyRows):
            # layer 0 of the fast path: gather + input dropout in one pass (or inside the aggregation kernel)
            feat_out = self._emit(ops.sage_dense(feat_in, adj_norm, self.f_lin_self, self.f_lin_neigh, self.act_name,
                                                 self.scale, self.offset, in_dropout=self._in_p(), chain_next=self.chain_next, roots_only=self.roots_only, pool_only=self.pool_only,
                                                 **self._drop_kw()))
            return feat_out, adj_norm, True, 0.
        feat_in = self.in_dropout(feat_in)
        if fused:
            # aggregate + both Linears + act/norm/add as one autograd node (single K = 2F input-gradient GEMM)
            feat_out = self._emit(ops.sage_dense(feat_in, adj_norm, self.f_lin_self, self.f_lin_neigh, self.act_name,
                                                 self.scale, self.offset, chain_next=self.chain_next, roots_only=self.roots_only, pool_only=self.pool_only, **self._drop_kw()))
        else:
            feat_neigh = self.spmm(adj_norm, feat_in)
            feat_out = self.f_lin_act_norm([feat_in, feat_neigh], [self.f_lin_self, self.f_lin_neigh],
                                           [self.act_name, self.act_name])
        return feat_out, adj_norm, True, 0.

    def complexity(self, dims_x, dims_adj):
        """One aggregation at the input width plus the self and neighbour Linears (shaDow/layers.py:486-494)."""
        if dims_x.num_nodes != dims_adj.num_nodes:
            raise ValueError("feature rows and adjacency rows differ")
        macs = (dims_adj.num_edges * dims_x.num_feats
                + sum(_dense_ops(dims_x.num_nodes, lin) for lin in (self.f_lin_self, self.f_lin_neigh)))
        return (Dims_X(dims_x.num_nodes, self.f_lin_self.out_features), Dims_adj(*dims_adj)), macs


def _respool_in_width(type_pool, type_res, dim, num_layers, task):
    """Input width of the read-out Linear (0 = no Linear at all).  Residue 'concat' stacks the layers side by
    side, every other residue keeps one width; a pooled read-out carries [root rows | pooled rows]."""
    stacked = type_res in ("cat", "concat")
    per_part = dim * num_layers if stacked else dim
    if type_pool == "center":
        if type_res == "none":
            return 0 if task == "node" else dim       # node task: the root's last-layer row goes out as it is
        return per_part
    return 2 * per_part


_RESIDUE = {
    "cat": lambda parts: torch.cat(parts, dim=1),
    "concat": lambda parts: torch.cat(parts, dim=1),
    "sum": lambda parts: torch.stack(parts, dim=0).sum(dim=0),
    "max": lambda parts: torch.stack(parts, dim=0).max(dim=0).values,
}


class ResPool(nn.Module):
    """Residue + pooling read-out (role of shaDow/layers.py:57-233).  'center' pooling is a row select;
    max / mean / sum pooling is one segment-reduction kernel over the subgraph row ranges.  Sort pooling
    (PyG global_sort_pool) is not provided.  Parameter names follow the reference's checkpoint layout:
    ``nn.1`` is the Linear, ``nn.2`` the activation module (it owns a tensor only for PReLU), ``scale`` /
    ``offset`` the feature normalisation."""
    POOLED = ("max", "mean", "sum")

    def __init__(self, dim_in, dim_out, num_layers, type_res, type_pool, dropout, act,
                 args_pool=None, prediction_task='node'):
        super().__init__()
        if type_pool != "center" and type_pool not in self.POOLED:
            raise NotImplementedError(f"pooling {type_pool!r} is not provided (sort pooling needs PyG)")
        self.type_pool, self.type_res, self.prediction_task = type_pool, type_res, prediction_task
        self.act_name = _check_act(act)
        self.dim_in = _respool_in_width(type_pool, type_res, dim_in, num_layers, prediction_task)
        self.dim_out = dim_out if self.dim_in > 0 else 0
        if self.dim_in > 0 and self.dim_out > 0:
            self.nn = nn.Sequential(nn.Dropout(p=dropout), nn.Linear(self.dim_in, self.dim_out, bias=True),
                                    _learned_act(act, self.dim_out) or nn.Identity())
            self.offset = nn.Parameter(torch.zeros(self.dim_out))
            self.scale = nn.Parameter(torch.ones(self.dim_out))

    def f_residue(self, feat_l):
        if self.type_res not in _RESIDUE:
            raise NotImplementedError(f"residue {self.type_res!r}")
        return _RESIDUE[self.type_res](list(feat_l))

    def aggr_target_emb(self, feat_src_dst):
        """Link task: the two end points' rows are multiplied; node task: identity."""
        if self.prediction_task == 'node':
            return feat_src_dst
        pairs = feat_src_dst.unflatten(0, (-1, 2))
        return pairs[:, 0] * pairs[:, 1]

    def _pool(self, feat, sizes_subg):
        # F.embedding_bag(arange(n), feat, offsets, mode) of the reference (layers.py:175,180) as one
        # segment-reduction kernel over the subgraph row ranges
        sizes = torch.as_tensor(sizes_subg, device=feat.device).reshape(-1)
        off = torch.zeros(sizes.numel() + 1, dtype=torch.int32, device=feat.device)
        off[1:] = torch.cumsum(sizes, dim=0)
        return ops.segment_pool(feat, off, self.type_pool)

    def forward(self, feats_in_l, idx_targets, sizes_subg):
        rows = torch.as_tensor(idx_targets, device=feats_in_l[-1].device).long()
        used = feats_in_l if self.type_res != 'none' else feats_in_l[-1:]       # residue 'none' reads the last layer
        merge = self.f_residue if self.type_res != 'none' else (lambda parts: parts[0])
        pooled = None
        if self.type_pool in self.POOLED and self.dim_in != 0 and all(f.is_cuda for f in used):
            # pooled read-out: every layer output is pooled AND read at the roots -- one node per output, one dense gradient
            sizes = torch.as_tensor(sizes_subg, device=rows.device).reshape(-1)
            off = torch.zeros(sizes.numel() + 1, dtype=torch.int32, device=rows.device)
            off[1:] = torch.cumsum(sizes, dim=0)
            pairs = [ops.pool_and_roots(f, off, rows, self.type_pool) for f in used]
            at_roots, pooled = merge([p[1] for p in pairs]), merge([p[0] for p in pairs])
        else:
            at_roots = merge([ops.select_roots(f, rows) for f in used])
        if self.dim_in == 0:
            return at_roots                                                    # layers.py:159-163
        feat_in = self.aggr_target_emb(at_roots)
        if self.type_pool in self.POOLED:
            if pooled is None:
                pooled = merge([self._pool(f, sizes_subg) for f in used])
            feat_in = torch.cat([feat_in, pooled], dim=1)
        # dropout -> Linear -> act -> norm (layers.py:110,114-118,199)
        drop, lin, act_mod = self.nn[0], self.nn[1], self.nn[2]
        if self.act_name in LEARNED_ACT:
            return ops.act_norm([act_mod(ops.linear(drop(feat_in), lin))], ['I'], self.scale, self.offset)
        return ops.linear_act_norm([drop(feat_in)], [lin], [self.act_name], self.scale, self.offset)


class GAT(shaDowLayer):
    """shaDow/layers.py:539-645.  Attention scores, edge softmax and the weighted
    aggregation of all heads run in fused HIP kernels (ops_gat)."""
    def __init__(self, dim_in, dim_out, dropout=0.0, act="relu", norm='norm_feat', mulhead=1, **kwargs):
        heads = int(mulhead)
        if heads < 1 or dim_out % heads:
            raise ValueError(f"GAT output width {dim_out} is not divisible into {mulhead} heads")
        self.mulhead, self.dim_slice = heads, dim_out // heads
        kwargs.pop('norm_dim', None)
        # every head slice of both branches (0: neighbours, 1: self -- the order GAT.forward normalises them in)
        # has its own scale / offset row
        super().__init__(dim_in, dim_out, dropout=dropout, act=act, norm=norm,
                         norm_dim=(2, heads, self.dim_slice), **kwargs)
        self.f_lin = nn.ModuleList([nn.Linear(dim_in, dim_out, bias=True), nn.Linear(dim_in, dim_out, bias=True)])  # self, neigh
        self.attention = nn.Parameter(nn.init.xavier_uniform_(torch.empty(2, heads, self.dim_slice)))

    def _adj_norm(self, adj, is_normed, device, dropedge=0):
        if not is_normed:
            csr = _as_device_csr(adj, device)
            # not normalised (data = 1,1,1...), only edge dropout (layers.py:590-596)
            return ops.NormAdj(csr, edge_w=ops.dropedge_mask(csr, dropedge))
        assert isinstance(adj, ops.NormAdj)
        return adj

    def forward_rows(self, feat_in, adj_norm, level):
        """The layer on the output rows of ``level`` only (tail.py).  The neighbour transform runs on every
        input row (attention reads it at the sources); the self transform, the normalisation and the edge
        kernels only see the level's rows (a square matrix with the other rows unconnected)."""
        from . import ops_gat
        feat_in = self.in_dropout(feat_in)
        adj_sq = level.square_adj(adj_norm)
        z_neigh = ops.linear(feat_in, self.f_lin[1])
        z_self_r = ops.linear(feat_in.index_select(0, level.self_idx), self.f_lin[0])
        if self.act is not None:
            z_neigh, z_self_r = self.act(z_neigh), self.act(z_self_r)
        z_self = torch.zeros(feat_in.shape[0], z_self_r.shape[1], dtype=z_self_r.dtype,
                             device=z_self_r.device).index_copy(0, level.self_idx, z_self_r)
        feat_neigh = ops_gat.gat_aggregate(adj_sq, z_self, z_neigh, self.attention, self.kact, self.mulhead)
        feat_neigh_r = feat_neigh.index_select(0, level.self_idx)
        if self.norm == 'norm_feat':
            return self._emit(ops.act_norm([feat_neigh_r, z_self_r], ['I', self.kact], self.scale, self.offset,
                                           seg=self.dim_slice, out_scale=0.5, **self._drop_kw()))
        return (feat_neigh_r + _torch_act(self.kact, z_self_r)) / 2

    def forward(self, inputs, sizes_subg):
        from . import ops_gat
        feat_in, adj, is_normed, dropedge = inputs
        adj_norm = self._adj_norm(adj, is_normed, feat_in.device, dropedge=dropedge)
        if isinstance(feat_in, ops.LazyRows):
            # layer 0 of the fast path: feat_full[node] and the layer's input dropout in ONE pass (line-padded rows, row maxima for the
            # paired Linear's operand scales) instead of gather + nn.Dropout + a row-maximum pass -- as GraphSAGE's layer 0 does
            feat_in, _seed = feat_in.gather_dropped(self._in_p())
        else:
            feat_in = self.in_dropout(feat_in)
        # (one launch for both transforms; when the fused tail below is their only consumer the same launch also leaves
        #  hn = act(z_neigh) -- in z_neigh's place -- and the attention's per-node terms, ops.GatPre)
        tail_only = self.norm == 'norm_feat' and self.act is None and ops_gat.gat_tail_usable(feat_in, self.f_lin[0].weight.shape[0], self.mulhead)
        pre = ops.GatPre(self.attention, ops.ACT_CODE[self.kact], self.mulhead) if tail_only else None
        z_self, z_neigh = ops.linear_pair(feat_in, self.f_lin[0], self.f_lin[1], gat=pre)
        if self.act is not None:
            z_self, z_neigh = self.act(z_self), self.act(z_neigh)
        # neigh branch: act -> per-head attention aggregate; both branches normalised per head slice
        if self.norm == 'norm_feat' and self.act is None:
            # (one autograd node for the aggregate and the normalisation when the fused kernels take the shape)
            res = ops_gat.gat_tail(adj_norm, z_self, z_neigh, self.attention, self.kact, self.mulhead, self.scale, self.offset,
                                   seg=self.dim_slice, out_scale=0.5, roots_only=self.roots_only or self.rows_next, **self._drop_kw())
            if res is not None:
                return self._emit(res), adj_norm, True, 0.
        feat_neigh = ops_gat.gat_aggregate(adj_norm, z_self, z_neigh, self.attention, self.kact,
                                           self.mulhead)
        if self.norm == 'norm_feat':
            # reference order: f_norm([neigh, self]) -> scale[0]=neigh, scale[1]=self (layers.py:620-622)
            feat_out = self._emit(ops.act_norm([feat_neigh, z_self], ['I', self.kact], self.scale, self.offset,
                                               seg=self.dim_slice, out_scale=0.5, **self._drop_kw()))
        else:
            feat_out = (feat_neigh + _torch_act(self.kact, z_self)) / 2
        return feat_out, adj_norm, True, 0.

    # per-edge cost model of the reference (shaDow/layers.py:628-645): 2 for the score sum, 20 for the softmax
    _EDGE_OPS_PER_HEAD = 2 + 20

    def complexity(self, dims_X, dims_adj):
        n, e = dims_X.num_nodes, dims_adj.num_edges
        width = self.f_lin[1].out_features
        macs = sum(_dense_ops(n, lin) + n * lin.out_features for lin in self.f_lin)      # Linears + attention dots
        macs += e * (self.mulhead * self._EDGE_OPS_PER_HEAD + width)                       # edge softmax + aggregation
        return (Dims_X(n, width), Dims_adj(*dims_adj)), macs
