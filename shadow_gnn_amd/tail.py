"""Target-only tail of the layer stack (an exact optimisation, off by default).

With residue 'none' and 'center' pooling on a node task the model reads ONE row per subgraph from the last
layer: the root's (shaDow/layers.py:159-163 -- ``feats_in_l[-1][idx_targets]``).  Row i of a GCN / GraphSAGE
layer depends on row i and on the in-subgraph neighbours of i of the layer below, nothing else (the feature
normalisation is per row, layers.py:329-338).  So the last layer is needed on the roots only, the layer below
on roots + their neighbours, ... until the needed set covers most of the batch (k-hop depth-2 subgraphs: after
two layers).  The reference computes every row of every layer; the rows dropped here never reach the loss, so
predictions and all parameter gradients are unchanged -- tests/test_tail_gpu.py checks both against the full
stack.

A pruned layer is a *rectangular* layer: r output rows, m_in input rows,

    out[k] = norm(act(lin_self(X[self_idx[k]]))) + norm(act(lin_neigh(sum_j A[rows[k], j] X[j])))      (SAGE)

on the rows of the batch adjacency selected by ``rows`` (same normalisation scales and drop-edge mask as the
full matrix: the scales are gathered, not recomputed).  The SpMM kernels are the ordinary CSR ones
(sl_spmm_csr_f32 takes any row count); the transposed matrix for the backward pass is built with a stable sort.

GAT transforms the neighbours before it aggregates (layers.py:604-611): its neighbour Linear runs on every input
row of the level, the attention / softmax / aggregation kernels on a square matrix over the input rows with only
the level's rows connected (``RectLevel.square``), and the self Linear and the normalisation on the level's
rows alone (GAT.forward_rows).
"""
from typing import List, Optional

import torch

from . import ops


class RectLevel:
    """One pruned layer: CSR of the selected rows with column ids in the numbering of its input."""

    def __init__(self, indptr, indices, edge_row, edge_pos, rows_full, in_ids_full, self_idx, m_in):
        self.indptr = indptr            # int32 [r + 1]
        self.indices = indices          # int32 [E], positions in the input tensor
        self.edge_row = edge_row        # int64 [E], local row of every edge
        self.edge_pos = edge_pos        # int64 [E], position of every edge in the full batch CSR
        self.rows_full = rows_full      # int64 [r], batch-level ids of the output rows
        self.in_ids_full = in_ids_full  # int64 [m_in] batch-level ids of the input rows, None = the whole batch
        self.self_idx = self_idx        # int64 [r], position of every output row in the input tensor
        self.m_in = int(m_in)
        self.r = int(rows_full.numel())
        self._t = None
        self._sq = None
        self.rows32 = self.in32 = None  # int32 copies of rows_full / in_ids_full (build_backward_levels)
        self.in_map32 = None            # int32 [n]: position of every batch row in in_ids_full, -1 = not an input (build_backward_levels)
        self.rows_ascending = False     # rows_full ascending: the edges are already in the square form's order (no sort, no bincount)

    @property
    def transposed(self):
        """(t_indptr, t_indices, t_perm) of the m_in x r transpose (stable sort by column)."""
        if self._t is None:
            cols = self.indices.long()
            perm = torch.argsort(cols, stable=True)
            t_indices = self.edge_row[perm].to(torch.int32)
            t_indptr = torch.zeros(self.m_in + 1, dtype=torch.int64, device=cols.device)
            if cols.numel():
                torch.cumsum(torch.bincount(cols, minlength=self.m_in), 0, out=t_indptr[1:])
            self._t = (t_indptr.to(torch.int32), t_indices, perm.to(torch.int32))
        return self._t

    def tensors(self):
        ts = [self.indptr, self.indices, self.edge_row, self.edge_pos, self.rows_full, self.self_idx]
        if self.in_ids_full is not None:
            ts.append(self.in_ids_full)
        ts.extend(x for x in (self.rows32, self.in32, self.in_map32) if x is not None)
        if self._t is not None:
            ts.extend(self._t)
        if self._sq is not None:
            csr, order = self._sq
            ts.extend([csr.indptr, csr.indices, order])
            if csr._t is not None:
                ts.extend(csr._t)
            if csr._edge_row is not None:
                ts.append(csr._edge_row)
        return ts

    @property
    def square(self):
        """(DeviceCSR, edge order) of the m_in x m_in matrix that keeps the edges of this level's rows and
        leaves every other row empty: layers without a rectangular form (GAT: the neighbour transform runs on
        the input rows anyway) run their ordinary kernels on it; rows outside the level produce values nobody
        reads."""
        if self._sq is None:
            if self.rows_ascending:
                # the selected rows ascend, so do their positions in the input numbering: the edges (grouped by selected row, in
                # row order) ARE the square form's edges in order -- row lengths scattered to those positions, no sort, no
                # bincount (a device-wide maximum read back by the host)
                dev = self.indptr.device
                lens = (self.indptr[1:] - self.indptr[:-1]).to(torch.int64)
                cnt = torch.zeros(self.m_in + 1, dtype=torch.int64, device=dev)
                cnt.index_copy_(0, self.self_idx + 1, lens)
                ip = torch.cumsum(cnt, 0)
                order = torch.arange(self.indices.numel(), device=dev)
                self._sq = (ops.DeviceCSR(ip.to(torch.int32), self.indices.contiguous()), order)
                return self._sq
            srow = self.self_idx[self.edge_row]                 # row of every edge in the input numbering
            order = torch.argsort(srow, stable=True)
            ip = torch.zeros(self.m_in + 1, dtype=torch.int64, device=srow.device)
            if srow.numel():
                torch.cumsum(torch.bincount(srow, minlength=self.m_in), 0, out=ip[1:])
            self._sq = (ops.DeviceCSR(ip.to(torch.int32), self.indices[order].contiguous()), order)
        return self._sq

    def square_adj(self, full: "ops.NormAdj") -> "ops.NormAdj":
        csr, order = self.square
        ew = full.edge_w[self.edge_pos[order]] if full.edge_w is not None else None
        rs = cs = None
        ids = self.in_ids_full
        if full.row_scale is not None:
            rs = full.row_scale if ids is None else full.row_scale[ids]
        if full.col_scale is not None:
            cs = full.col_scale if ids is None else full.col_scale[ids]
        return ops.NormAdj(csr, edge_w=ew, row_scale=rs, col_scale=cs)

    def norm(self, full: "ops.NormAdj"):
        """(edge_w, row_scale, col_scale) of this level, gathered from the normalised full adjacency."""
        ew = full.edge_w[self.edge_pos] if full.edge_w is not None else None
        rs = full.row_scale[self.rows_full] if full.row_scale is not None else None
        cs = full.col_scale
        if cs is not None and self.in_ids_full is not None:
            cs = cs[self.in_ids_full]
        return ew, rs, cs


# host seconds spent BLOCKED in the size read-backs of the row-set builders below (the stream they run on -- the extractor's
# prefetch stream -- also carries the sampler's kernels): MinibatchShallowExtractor adds the difference to its wait_s
_SYNC_WAIT = [0.0]


def _select_rows(indptr: torch.Tensor, rows: torch.Tensor):
    """CSR row selection: local indptr (int64), local row of every edge, position of every edge in the source."""
    dev = indptr.device
    start = indptr[rows].long()
    lens = indptr[rows + 1].long() - start
    ip = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=ip[1:])
    import time as _time
    t0 = _time.perf_counter()
    E = int(ip[-1])                                    # (host sync: the edge count sizes the arrays below)
    _SYNC_WAIT[0] += _time.perf_counter() - t0
    er = torch.repeat_interleave(torch.arange(rows.numel(), device=dev), lens, output_size=E)
    pos = start[er] + (torch.arange(E, device=dev) - ip[er])
    return ip, er, pos


def build_tail_plan(csr: "ops.DeviceCSR", targets: torch.Tensor, num_layers: int, frac: float = 0.5,
                    eager_transpose: bool = False) -> List[RectLevel]:
    """Levels for the LAST len(result) layers, bottom (first to run) first.  Empty: nothing worth pruning.
    A level is kept only while its output rows are at most ``frac`` of the batch; the lowest level always reads
    the full-size tensor of the layer below (or the input features).
    Two host syncs per level (array sizes): MinibatchShallowExtractor builds the plan on its prefetch stream,
    next to the sampler's own sync, so the training stream never waits for it."""
    n = csr.n
    dev = csr.device
    rows = torch.as_tensor(targets, device=dev).long().reshape(-1)
    levels: List[RectLevel] = []
    for layer in reversed(range(num_layers)):
        if rows.numel() > frac * n:
            break
        ip, er, pos = _select_rows(csr.indptr, rows)
        cols = csr.indices[pos].long()
        mask = torch.zeros(n, dtype=torch.bool, device=dev)
        mask.index_fill_(0, rows, True)
        mask.index_fill_(0, cols, True)
        cnt = int(mask.sum())                          # (host sync)
        if layer == 0 or cnt > frac * n:               # input = the whole batch, batch numbering
            levels.append(RectLevel(ip.to(torch.int32), cols.to(torch.int32), er, pos, rows, None, rows, n))
            break
        in_ids = mask.nonzero().reshape(-1)
        newid = torch.cumsum(mask, 0) - 1
        levels.append(RectLevel(ip.to(torch.int32), newid[cols].to(torch.int32), er, pos, rows, in_ids, newid[rows], cnt))
        rows = in_ids
    levels.reverse()
    if eager_transpose:
        for lv in levels:
            if eager_transpose == "square":
                lv.square[0].transposed
            else:
                lv.transposed
    return levels


class _RectGatherSpMM(torch.autograd.Function):
    """(X[self_idx], A_rect X) with one dense input gradient: dX = A_rect^T dAX, then += dXs at self_idx."""

    @staticmethod
    def forward(ctx, X, level: RectLevel, ew, rs, cs):
        X = X.contiguous().float()
        ctx.level, ctx.norm = level, (ew, rs, cs)
        ctx.set_materialize_grads(False)       # (GCN never uses the self rows: their gradient stays None)
        ctx.m_in = X.shape[0]
        assert X.shape[0] == level.m_in, (X.shape, level.m_in)
        AX = ops._spmm_raw(level.indptr, level.indices, ew, None, rs, cs, X, level.r)
        Xs = X.index_select(0, level.self_idx)
        return Xs, AX

    @staticmethod
    def backward(ctx, dXs, dAX):
        level = ctx.level
        ew, rs, cs = ctx.norm
        ti, tx, tp = level.transposed
        if dAX is None and dXs is None:
            return None, None, None, None, None
        if dAX is None:
            dX = torch.zeros(ctx.m_in, dXs.shape[1], dtype=torch.float32, device=dXs.device)
        else:
            # (diag(rs) W diag(cs))^T = diag(cs) W^T diag(rs)
            dX = ops._spmm_raw(ti, tx, ew, tp if ew is not None else None, cs, rs, dAX.contiguous().float(), level.m_in)
        if dXs is not None:
            dX.index_add_(0, level.self_idx, dXs.float())
        return dX, None, None, None, None


def rect_gather_spmm(X: torch.Tensor, level: RectLevel, full_adj: "ops.NormAdj"):
    """Returns (X[level.self_idx], A[level rows, :] @ X) under the normalisation of ``full_adj``."""
    ew, rs, cs = level.norm(full_adj)
    return _RectGatherSpMM.apply(X, level, ew, rs, cs)


class TopBackwardPlan:
    """Row sets of the EXACT row-sparse backward pass of the top GraphSAGE layer (ops._SageDense._sparse_top_backward).

    Residue 'none' + centre pooling on a node task reads one row per subgraph of the last layer's output
    (shaDow/layers.py:159-163), so the gradient of that output is zero outside the roots R, the top layer's dZs / dZn are
    zero outside R, and its input gradient  dX = dZs Ws + A^T (dZn Wn)  is zero outside  T = R u N(R)  (the roots and their
    in-subgraph neighbours: ~7 % of a depth-2 k-hop batch).  The forward pass is untouched -- every row of every layer is
    computed, as in the reference; the backward pass multiplies by the zeros the reference's autograd multiplies by, or skips
    them: the same gradients.  Built by two small kernels (sl_top_plan) and ONE host sync (the size of T): the minibatch
    extractor builds it on its prefetch stream.  ``ok`` False: a root row lists a neighbour twice (a multigraph) -- the
    dense pass is taken."""

    def __init__(self, csr: "ops.DeviceCSR", targets: torch.Tensor, want_filter: bool = True):
        import ctypes as C

        from . import _lib
        n, dev = csr.n, csr.device
        tg = torch.as_tensor(targets, device=dev).reshape(-1)
        self.targets32 = (tg if tg.dtype == torch.int32 else tg.to(torch.int32)).contiguous()
        self.rows64 = self.targets32.long()
        P = int(self.targets32.numel())
        cap = n + P
        i32 = dict(dtype=torch.int32, device=dev)
        off = torch.zeros(P + 3, **i32)
        T, slot, epos, self_idx = torch.empty(cap, **i32), torch.empty(cap, **i32), torch.empty(cap, **i32), torch.empty(max(P, 1), **i32)
        lib = _lib.load()
        st = ops._stream(csr.indptr)
        _lib.check(lib.sl_top_plan(csr.indptr.data_ptr(), csr.indices.data_ptr(), self.targets32.data_ptr(), P, cap, off.data_ptr(),
                                   T.data_ptr(), slot.data_ptr(), epos.data_ptr(), self_idx.data_ptr(), st))
        # batch row -> its position in T, or all ones (any value >= t stands for "the zero row behind a [t, F] compact tensor":
        # the row map of sl_spmm_blockdiag_rows_f32 / sl_gemm_an_bwd_corr), and -- round 5 -- the transposed adjacency restricted
        # to the columns T: A^T dZn of the layer below the row-sparse pass has terms from the rows T only, its aggregation walks
        # ~300 of a subgraph's ~2 000 transposed edges (ops._SageDense._compact_dz_backward).  Built by the same C call chain
        # on this stream: still ONE host sync per plan.
        self.rowmap = torch.empty(n, **i32)
        self.f_indptr = self.f_indices = self.f_perm = None
        self._t_keep = ()
        # (``want_filter`` False: nobody reads the filtered structure -- ops._at_dzn_on_rows takes it for 128 < F <= 256 only,
        #  MinibatchShallowExtractor.attach_model knows the model's widths -- and the six kernels + three [e] tensors are not spent)
        filt = bool(want_filter and P and csr.e > 0)
        if filt:
            ti, tx, tp = csr.transposed
            self.f_indptr = torch.empty(n + 1, **i32)
            self.f_indices, self.f_perm = torch.empty(csr.e, **i32), torch.empty(csr.e, **i32)
            work = torch.empty(n // 1024 + 8, **i32)
            _lib.check(lib.sl_top_plan_filter(T.data_ptr(), off.data_ptr(), P, cap, ti.data_ptr(), tx.data_ptr(), tp.data_ptr(), n,
                                              self.rowmap.data_ptr(), self.f_indptr.data_ptr(), self.f_indices.data_ptr(),
                                              self.f_perm.data_ptr(), work.data_ptr(), st))
            self._t_keep = (ti, tx, tp, work)                     # (built on this stream: handed to the consumer's stream with the plan)
        else:
            self.rowmap.fill_(-1)
        import time as _time
        t0 = _time.perf_counter()
        t, bad, ef = (int(x) for x in off[P:P + 3].tolist()) if P else (0, 0, 0)          # (the one host sync)
        self.sync_wait_s = _time.perf_counter() - t0                # (host time BLOCKED on this stream, not busy: minibatch.wait_s)
        self.ok = bool(P > 0 and bad == 0 and t <= cap)
        self.t = t if self.ok else 0
        self.f_nnz = ef if self.ok else 0
        if not self.ok:
            self.f_indptr = self.f_indices = self.f_perm = None
        self.T32, self.slot, self.epos, self.self_idx = T[:self.t], slot[:self.t], epos[:self.t], self_idx[:P]
        if not filt and self.ok:                      # (no filter pass -- not wanted, or a batch without a single edge: the row map alone)
            self.rowmap[self.T32.long()] = torch.arange(self.t, **i32)
        self.n = n
        self.num_roots = P
        self._indptr_ptr = csr.indptr.data_ptr()

    def matches(self, csr: "ops.DeviceCSR", num_roots: int) -> bool:
        return self.ok and csr.n == self.n and csr.indptr.data_ptr() == self._indptr_ptr and num_roots == self.num_roots

    def tensors(self):
        extra = [x for x in (self.f_indptr, self.f_indices, self.f_perm, *self._t_keep) if x is not None]
        return [self.targets32, self.rows64, self.T32, self.slot, self.epos, self.self_idx, self.rowmap, *extra]


def build_backward_levels(csr: "ops.DeviceCSR", targets: torch.Tensor, max_levels: int = 2, frac: float = 0.25,
                          targets_ascending: bool = False) -> List[RectLevel]:
    """Nested row sets of a row-sparse BACKWARD pass, top layer first: level 0 = the roots' rows with their inputs T = R u N(R),
    level 1 = the rows T with inputs T2 = T u N(T), ... -- every level compact (column ids renumbered into its input set), kept
    while that input set is at most ``frac`` of the batch.  Depth-2 k-hop batches stop after one level (T2 is the whole
    subgraph), depth-3 batches after two (T: 0.5 %, T2: 10 % of the rows).  The square form and its transpose are built here
    (GAT's attention backward runs its ordinary kernels on them, ops_gat._GatTail).  Two host syncs per level: the extractor
    calls this on its prefetch stream.  ``targets_ascending``: the caller's word that ``targets`` ascend (a collated batch's roots
    do: one per subgraph, subgraphs in order) -- the levels' square forms then need no sort (the lower levels' rows ascend by
    construction)."""
    n, dev = csr.n, csr.device
    rows = torch.as_tensor(targets, device=dev).long().reshape(-1)
    levels: List[RectLevel] = []
    ascending = bool(targets_ascending)
    for li in range(max_levels):
        ip, er, pos = _select_rows(csr.indptr, rows)
        cols = csr.indices[pos].long()
        mask = torch.zeros(n, dtype=torch.bool, device=dev)
        mask.index_fill_(0, rows, True)                     # (`mask[rows] = True` stages its scalar through a blocking H2D copy)
        mask.index_fill_(0, cols, True)
        if li == 0 and ascending and rows.numel() > 1:
            # the caller's word is checked on the device and rides on the read-back below: targets that do not strictly ascend
            # (a collate / cache path that reorders or repeats roots) fill the mask -- the input set is then the whole batch, no
            # level is kept and the dense backward pass runs (RectLevel.square's sort-free form would be wrong for them)
            mask.logical_or_((rows[1:] <= rows[:-1]).any())
        import time as _time
        t0 = _time.perf_counter()
        in_ids = mask.nonzero().reshape(-1)                 # (host sync) ascending
        _SYNC_WAIT[0] += _time.perf_counter() - t0
        if in_ids.numel() > frac * n:
            break
        newid = torch.cumsum(mask, 0) - 1
        lv = RectLevel(ip.to(torch.int32), newid[cols].to(torch.int32), er, pos, rows, in_ids, newid[rows], in_ids.numel())
        lv.rows_ascending = ascending
        ascending = True                                    # (in_ids come out of nonzero(): ascending)
        lv.square[0].transposed
        lv.rows32 = rows.to(torch.int32)
        lv.in32 = in_ids.to(torch.int32)
        # (the layer BELOW the last level receives its output gradient on the rows in_ids: with this map its attention backward
        #  reads the compact gradient in place, ops_gat.MAP_ROWS_GRADIENT)
        lv.in_map32 = torch.where(mask, newid, newid.new_full((), -1)).to(torch.int32)
        levels.append(lv)
        rows = in_ids
    if levels:
        csr.transposed                                      # (the dense layers' column walk: built here, off the training stream)
    return levels
