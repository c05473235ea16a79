"""torch front-end of the sl_* layer kernels (libshadow_hip.so).

``DeviceCSR`` is the device-resident stand-in for the scipy CSR matrix the
reference hands to layer 0 (shaDow/minibatch.py:468, shaDow/layers.py:427,467);
``NormAdj`` is what layer 0 returns as ``adj_norm`` and later layers receive
(the reference threads a torch sparse COO tensor, layers.py:436,484,626).

Every op launches hand-written HIP kernels through the C ABI on the current
torch stream.  No CPU / torch fallback: tensors must live on a ROCm device.
"""
import ctypes as C
import os
import sys
from typing import NamedTuple, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check

ACT_CODE = {"I": 0, "relu": 1, "elu": 2, "tanh": 3, "leakyrelu": 4}


class KernelTimer:
    """Optional live timing of the hand-written kernels with HIP events on the
    stream they are launched on (bench.py's roofline measurement).  Each record:
    kernel class -> [events..., algorithmic bytes]."""
    active = None

    def __init__(self):
        self.rec = {}
        self.count = {}       # launches per kernel class (all of them; only the first MAX_PER_KERNEL carry events)

    def __enter__(self):
        KernelTimer.active = self
        _lib.load().sl_prof_enable(1)       # the one-call layer entries time their own kernels (csrc/prof.hip)
        return self

    def __exit__(self, *a):
        KernelTimer.active = None
        _lib.load().sl_prof_enable(0)

    def _c_records(self):
        """(name, launches, timed, total ms, bytes, flops) of the kernels launched inside the C layer entries."""
        lib = _lib.load()
        need = int(lib.sl_prof_dump(None, 0))
        buf = C.create_string_buffer(need + 16)
        lib.sl_prof_dump(buf, need + 16)
        rows = []
        for line in buf.value.decode().splitlines():
            f = line.split("\t")
            if len(f) == 6:
                rows.append((f[0], int(f[1]), int(f[2]), float(f[3]), float(f[4]), float(f[5])))
        return rows

    def summary(self):
        # merge the Python-side event pairs (direct primitive calls) with the C-side ones (one-call entries)
        acc = {}
        for name, items in self.rec.items():
            a = acc.setdefault(name, [0, 0, 0.0, 0.0, 0.0])
            a[0] += max(len(items), self.count.get(name, 0)); a[1] += len(items)
            a[2] += sum(it[0].elapsed_time(it[1]) for it in items)
            a[3] += sum(it[2] for it in items); a[4] += sum(it[3] for it in items)
        for name, launches, timed, ms, by, fl in self._c_records():
            a = acc.setdefault(name, [0, 0, 0.0, 0.0, 0.0])
            a[0] += launches; a[1] += timed; a[2] += ms; a[3] += by; a[4] += fl
        out = {}
        for name, (n, k, ms, by, fl) in acc.items():
            k = max(1, k)
            out[name] = dict(launches=max(n, k), timed_launches=k, total_ms=ms / k * max(n, k), avg_ms=ms / k,
                             bytes_per_launch=by / k, gbps=(by / 1e9) / (ms / 1e3) if ms > 0 else 0.0,
                             flops_per_launch=fl / k)
        return out


class _timed:
    MAX_PER_KERNEL = 512          # long runs: keep the number of live HIP events bounded

    def __init__(self, name, nbytes, device, flops=0):
        self.t = KernelTimer.active
        if self.t is not None:
            self.t.count[name] = self.t.count.get(name, 0) + 1
            if len(self.t.rec.get(name, ())) >= _timed.MAX_PER_KERNEL:
                self.t = None
        if self.t is not None:
            self.name, self.nbytes, self.flops = name, nbytes, flops
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.t is not None:
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.t is not None:
            self.e1.record()
            self.t.rec.setdefault(self.name, []).append((self.e0, self.e1, self.nbytes, self.flops))


def _stream(t: torch.Tensor):
    # (the raw handle of torch's current stream on t's device: ~20x cheaper than building a torch.cuda.Stream object,
    #  and every op launch needs it)
    return torch._C._cuda_getCurrentRawStream(t.device.index if t.device.index is not None else torch.cuda.current_device())


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("shadow_gnn_amd ops need tensors on a ROCm device (no CPU fallback)")


def placeholder(n: int, F: int, device) -> torch.Tensor:
    """An [n, F] fp32 tensor with ONE float of storage behind it (strides (0, 0) -- also for n == 1 or F == 1, where expand() would
    keep a unit stride): what autograd carries between two nodes whose real hand-over travels on a link object.  The consumer
    recognises it by its data pointer and strides."""
    return torch.empty(1, dtype=torch.float32, device=device).as_strided((int(n), int(F)), (0, 0))


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    if t.stride(-1) != 1 or (t.dim() == 2 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


class DeviceCSR:
    """Square batch adjacency (block diagonal) in HBM: int32 tensors holding
    uint32 values, all-ones data (the sampler emits data = 1., .cpp:411,423)."""

    def __init__(self, indptr: torch.Tensor, indices: torch.Tensor, subg_off: Optional[torch.Tensor] = None,
                 subg_edge_off: Optional[torch.Tensor] = None, max_subg_nodes: int = 0, row_entries_bound: int = 0):
        """``subg_off`` / ``subg_edge_off`` ([P+1] int32 node / edge offsets of the diagonal blocks, as
        the sampler returns them) let SpMM stage each subgraph's features in LDS
        (sl_spmm_blockdiag_f32).  ``row_entries_bound``: an upper bound of a row's entries when the caller knows one (the k of a
        top-k PPR batch, budget + 1 of a budgeted k-hop batch; 0: unknown) -- batches with long rows aggregate their 256-float
        rows on the pipelined CSR kernel (sl_set_spmm_wide_pipe)."""
        _need_cuda(indptr, indices)
        assert indptr.dtype == torch.int32 and indices.dtype == torch.int32
        self.indptr = indptr.contiguous()
        self.indices = indices.contiguous()
        self.n = int(indptr.numel()) - 1
        self.e = int(indices.numel())
        self._edge_row = None
        self._t = None
        self._blocks = None
        self.subg_off = subg_off.contiguous() if subg_off is not None else None
        self.subg_edge_off = subg_edge_off.contiguous() if subg_edge_off is not None else None
        self.max_subg_nodes = int(max_subg_nodes)
        self.row_entries_bound = int(row_entries_bound)
        if self.subg_off is not None:
            assert self.subg_edge_off is not None and self.subg_edge_off.numel() == self.subg_off.numel()
            assert self.subg_off.dtype == torch.int32 and self.subg_off.is_cuda
            assert self.subg_edge_off.dtype == torch.int32 and self.subg_edge_off.is_cuda

    @property
    def spmm_blocks(self):
        """(node offsets, edge offsets, largest block) the block-diagonal SpMM stages in LDS: the subgraphs themselves, or --
        small subgraphs (PPR: ~150 rows each, a 128-row kernel pass and a bit) -- consecutive subgraphs joined into groups
        of up to 384 rows / 1024 edges (sl_merge_subgraphs; computed once per batch, a group is still a diagonal block)."""
        if self.subg_off is None:
            return (None, None, 0)
        if self._blocks is None:
            P = int(self.subg_off.numel()) - 1
            self._blocks = (self.subg_off, self.subg_edge_off, self.max_subg_nodes)
            if MERGE_SMALL_SUBGRAPHS and 2 <= P <= 8191 and self.n < MERGE_BELOW_AVG_ROWS * P:
                goff, geoff = torch.empty_like(self.subg_off), torch.empty_like(self.subg_edge_off)
                check(_lib.load().sl_merge_subgraphs(self.subg_off.data_ptr(), self.subg_edge_off.data_ptr(), P, 384, 1024, goff.data_ptr(),
                                                     geoff.data_ptr(), _stream(self.indptr)))
                self._blocks = (goff, geoff, max(384, self.max_subg_nodes))
        return self._blocks

    @property
    def shape(self):
        return (self.n, self.n)

    @property
    def size(self):          # scipy's nnz-like attribute used by calc_complexity_step
        return self.e

    @property
    def device(self):
        return self.indptr.device

    @classmethod
    def from_scipy(cls, adj, device):
        """Upload a scipy CSR (the reference's layer-0 input type)."""
        ip = torch.from_numpy(np.ascontiguousarray(adj.indptr, dtype=np.int64).astype(np.int32))
        ix = torch.from_numpy(np.ascontiguousarray(adj.indices, dtype=np.int64).astype(np.int32))
        if adj.data.size and not np.all(adj.data == 1):
            raise ValueError("DeviceCSR carries binary adjacencies only (graph_utils.py:83,111)")
        return cls(ip.to(device), ix.to(device))

    @property
    def edge_row(self) -> torch.Tensor:
        if self._edge_row is None:
            er = torch.empty(max(1, self.e), dtype=torch.int32, device=self.device)[:self.e]
            check(_lib.load().sl_csr_edge_rows(self.indptr.data_ptr(), self.n, self.e, er.data_ptr(),
                                               _stream(er)))
            self._edge_row = er
        return self._edge_row

    @property
    def transposed(self):
        """(t_indptr, t_indices, t_perm) of A^T, built once per batch."""
        if self._t is None:
            dev = self.device
            ti = torch.empty(self.n + 1, dtype=torch.int32, device=dev)
            tx = torch.empty(max(1, self.e), dtype=torch.int32, device=dev)[:self.e]
            tp = torch.empty(max(1, self.e), dtype=torch.int32, device=dev)[:self.e]
            work = torch.empty(self.n + 2 * self.e + 16, dtype=torch.int32, device=dev)
            check(_lib.load().sl_csr_transpose(self.indptr.data_ptr(), self.indices.data_ptr(),
                                               self.edge_row.data_ptr(), self.n, self.e, ti.data_ptr(),
                                               tx.data_ptr(), tp.data_ptr(), work.data_ptr(), _stream(ti)))
            self._t = (ti, tx, tp)
        return self._t


class NormAdj:
    """Normalised adjacency  diag(row_scale) (A o edge_w) diag(col_scale)."""

    def __init__(self, csr: DeviceCSR, edge_w: Optional[torch.Tensor] = None,
                 row_scale: Optional[torch.Tensor] = None, col_scale: Optional[torch.Tensor] = None):
        self.csr = csr
        self.edge_w = edge_w
        self.row_scale = row_scale
        self.col_scale = col_scale

    @property
    def shape(self):
        return self.csr.shape

    def to_dense(self) -> torch.Tensor:
        """Dense fp32 copy (tests / debugging only)."""
        n = self.csr.n
        rows = self.csr.edge_row.long()
        cols = self.csr.indices.long()
        w = torch.ones(self.csr.e, device=self.csr.device) if self.edge_w is None else self.edge_w.clone()
        if self.row_scale is not None:
            w = w * self.row_scale[rows]
        if self.col_scale is not None:
            w = w * self.col_scale[cols]
        d = torch.zeros(n, n, device=self.csr.device)
        d.index_put_((rows, cols), w, accumulate=True)
        return d


def rows_multi(jobs, idx: Optional[torch.Tensor], rows: int):
    """Several row copies under one int64 row index in ONE launch (sl_rows_multi).  ``jobs``: (mode, src, dst) with mode
    'gather' (dst[i] = src[idx[i]]), 'clear' (dst[i] = 0, src None) or 'scatter' (dst[idx[i]] = src[i]); fp32, unit column stride."""
    if rows == 0 or not jobs:
        return
    arr = (_lib.SlRowsJob * len(jobs))()
    for q, (mode, src, dst) in enumerate(jobs):
        assert dst.dtype == torch.float32 and dst.dim() == 2 and dst.stride(1) == 1
        assert src is None or (src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1 and src.shape[1] == dst.shape[1])
        arr[q].mode = {"gather": 0, "clear": 1, "scatter": 2}[mode]
        arr[q].src, arr[q].lds = (src.data_ptr(), src.stride(0)) if src is not None else (None, 0)
        arr[q].dst, arr[q].ldd, arr[q].width = dst.data_ptr(), dst.stride(0), dst.shape[1]
    assert idx is None or (idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() >= rows)
    check(_lib.load().sl_rows_multi(arr, len(jobs), idx.data_ptr() if idx is not None else None, int(rows), _stream(jobs[0][2])))


def degree_scales(csr: DeviceCSR, edge_w: Optional[torch.Tensor], mode: str) -> torch.Tensor:
    out = torch.empty(max(1, csr.n), dtype=torch.float32, device=csr.device)[:csr.n]
    check(_lib.load().sl_degree_scales(csr.indptr.data_ptr(), edge_w.data_ptr() if edge_w is not None else None,
                                       csr.n, {"rw": 0, "sym": 1}[mode], out.data_ptr(), _stream(out)))
    return out


def dropedge_mask(csr: DeviceCSR, dropedge: float, symmetric: bool = False) -> Optional[torch.Tensor]:
    """Edge keep-mask of the reference's drop-edge: int(nnz*p) positions drawn WITH
    replacement are zeroed (graph_utils.py:85-88, layers.py:592-596); the symmetric
    variant keeps an edge only when its mate survived too (graph_utils.py:114-123)."""
    if dropedge <= 0 or csr.e == 0:
        return None
    num = int(csr.e * dropedge)
    m = torch.ones(csr.e, dtype=torch.float32, device=csr.device)
    if num > 0:
        # (index_fill_ with a scalar: `m[idx] = 0` would stage a host scalar through a blocking H2D copy every step)
        m.index_fill_(0, torch.randint(0, csr.e, (num,), device=csr.device), 0.0)
    if symmetric:
        _ti, _tx, tp = csr.transposed
        m = m * m[tp.long()]
    return m


def adj_norm_rw(csr: DeviceCSR, dropedge: float = 0.0) -> NormAdj:
    """D^-1 A with drop-edge (frontend/graph_utils.py:67-95, torch-sparse branch)."""
    m = dropedge_mask(csr, dropedge)
    return NormAdj(csr, edge_w=m, row_scale=degree_scales(csr, m, "rw"))


def adj_norm_sym(csr: DeviceCSR, dropedge: float = 0.0) -> NormAdj:
    """D^-1/2 A D^-1/2 with symmetric drop-edge (frontend/graph_utils.py:109-145)."""
    m = dropedge_mask(csr, dropedge, symmetric=True)
    s = degree_scales(csr, m, "sym")
    return NormAdj(csr, edge_w=m, row_scale=s, col_scale=s)


BLOCKDIAG_MIN_F = 96      # below this width the per-edge gather kernels are faster (measured)
MERGE_SMALL_SUBGRAPHS = True
MERGE_BELOW_AVG_ROWS = 190   # two average subgraphs must fit the 384-row tile


def _adj_struct(adj: "NormAdj", need_transpose: bool):
    """The ctypes image of a NormAdj for the one-call layer entries (the tensors stay owned by ``adj``)."""
    # (one image per adjacency and form: a layer stack passes the same NormAdj to ten entries per step)
    cache = adj.__dict__.setdefault("_struct_cache", {})
    key = (bool(need_transpose), id(adj.edge_w), id(adj.row_scale), id(adj.col_scale))
    hit = cache.get(key)
    if hit is not None:
        return hit
    c = adj.csr
    ptr = lambda t: t.data_ptr() if t is not None else None
    ti = tx = tp = None
    if need_transpose:
        ti, tx, tp = c.transposed
    boff, beoff, bmax = c.spmm_blocks
    st = _lib.SlNormAdj(c.indptr.data_ptr(), c.indices.data_ptr(), ptr(adj.edge_w), ptr(adj.row_scale), ptr(adj.col_scale),
                        ptr(ti), ptr(tx), ptr(tp), ptr(boff), ptr(beoff),
                        (int(boff.numel()) - 1) if boff is not None else 0, bmax, c.n, c.e, int(getattr(c, "row_entries_bound", 0)))
    cache[key] = st
    return st


# One C call per GraphSAGE layer pass (sl_sage_fwd / sl_sage_bwd_chain) instead of one per kernel.  A KernelTimer does
# not change the path: the C entries time their own kernels (csrc/prof.hip).
FUSED_LAYER_CALLS = True


def _spmm_raw(indptr, indices, edge_w, edge_perm, row_scale, col_scale, X, n, blocks=None, out=None, row_entries_bound=0):
    """``blocks`` = (subg_off, subg_edge_off, max_subg_nodes) of a block-diagonal adjacency.
    ``out``: optional [n, F] destination (may be a column slice of a wider buffer).  ``row_entries_bound``: DeviceCSR's."""
    X = _f32c(X)
    F = X.shape[1]
    if out is not None:
        Y = out
    elif X.stride(0) != F and X.stride(0) % 32 == 0:
        # line-padded input rows (LazyRows.gather_dropped): keep the product's rows on the same aligned pitch
        Y = torch.empty(n, X.stride(0), dtype=torch.float32, device=X.device)[:, :F]
    else:
        Y = torch.empty(n, F, dtype=torch.float32, device=X.device)
    e = int(indices.numel())
    # algorithmic bytes: indptr + indices (+ edge values) + read X + write A.X
    nbytes = 4 * (n + 1) + 4 * e + (4 * e if edge_w is not None else 0) + 8 * n * F
    lib = _lib.load()
    opt = lambda t: t.data_ptr() if t is not None else None
    with _timed(f"spmm_F{F}", nbytes, X.device):
        # (rows wider than 128 floats: the pipelined CSR kernel -- a row per wavefront, gathers out of the L2 -- as inside the
        #  sl_sage_* / sl_gcn_* entries, sl_set_spmm_wide_pipe; the block-diagonal LDS kernel for 96 .. 128 floats)
        mode = lib.sl_set_spmm_wide_pipe(-1)
        wide = (128 < F <= 256 and F % 4 == 0 and X.stride(0) % 4 == 0 and Y.stride(0) % 4 == 0 and X.data_ptr() % 16 == 0
                and Y.data_ptr() % 16 == 0 and (mode == 2 or (mode == 1 and row_entries_bound > 64 and n >= 98304)))   # (layer_fused.hip: spmm_wide_pipe)
        if blocks is not None and blocks[0] is not None and F >= BLOCKDIAG_MIN_F and not wide:
            off, eoff, mn = blocks
            check(lib.sl_spmm_blockdiag_f32(
                indptr.data_ptr(), indices.data_ptr(), opt(edge_w), opt(edge_perm), opt(row_scale), opt(col_scale),
                X.data_ptr(), X.stride(0), Y.data_ptr(), Y.stride(0), n, F, off.data_ptr(), eoff.data_ptr(),
                int(off.numel()) - 1, mn, None, _stream(X)))
        else:
            check(lib.sl_spmm_csr_f32(
                indptr.data_ptr(), indices.data_ptr(), opt(edge_w), opt(edge_perm), opt(row_scale), opt(col_scale),
                X.data_ptr(), X.stride(0), Y.data_ptr(), Y.stride(0), n, F, _stream(X)))
    return Y


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, adj: NormAdj):
        _need_cuda(X)
        ctx.adj = adj
        c = adj.csr
        return _spmm_raw(c.indptr, c.indices, adj.edge_w, None, adj.row_scale, adj.col_scale, X, c.n,
                         c.spmm_blocks, row_entries_bound=c.row_entries_bound)

    @staticmethod
    def backward(ctx, dY):
        adj = ctx.adj
        ti, tx, tp = adj.csr.transposed
        # (diag(rs) W diag(cs))^T = diag(cs) W^T diag(rs)
        c = adj.csr          # the transpose of a block-diagonal matrix has the same blocks
        dX = _spmm_raw(ti, tx, adj.edge_w, tp if adj.edge_w is not None else None, adj.col_scale,
                       adj.row_scale, dY, c.n, c.spmm_blocks, row_entries_bound=c.row_entries_bound)
        return dX, None


def spmm(adj: NormAdj, X: torch.Tensor) -> torch.Tensor:
    """adj @ X  (torch.sparse.mm(adj_norm, X), shaDow/layers.py:326-327)."""
    out = _SpMM.apply(X, adj)
    fire_deferred()
    return out


def gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """table[idx]  (feat_full[subgs.node], shaDow/minibatch.py:469)."""
    _need_cuda(table, idx)
    assert table.dim() == 2 and idx.dtype == torch.int32 and table.dtype == torch.float32
    n, F = int(idx.numel()), int(table.shape[1])
    out = torch.empty(n, F, dtype=torch.float32, device=table.device)
    with _timed(f"gather_F{F}", 8 * n * F + 4 * n, table.device):
        check(_lib.load().sl_gather_rows_f32(table.data_ptr(), table.stride(0), idx.data_ptr(), n, F,
                                             out.data_ptr(), out.stride(0), _stream(out)))
    return out


# Layer 0 of the fast path reads its input as LazyRows.  Two ways to do that were built and measured (MI355X, products
# shape, 1024 roots, F0 = 100; scripts/probe_spmm_gather.py): the gather (+ input dropout) INSIDE the block-diagonal
# SpMM (sl_spmm_blockdiag_gather_f32: 152 us incl. the dense copy the self Linear needs) and ONE gather + dropout pass
# into 128-byte-padded rows followed by the dense SpMM (43 + 81 us).  The dependent id -> row loads inside the SpMM's
# software pipeline cost more than the extra pass saves, so the second is the default.
FUSE_GATHER_INTO_SPMM = False


class LazyRows:
    """table[idx] that has not been gathered yet: the fast minibatch path hands the model this instead of the
    gathered feature matrix, so that layer 0 reads ``feat_full[subgs.node]`` (shaDow/minibatch.py:469) directly
    inside its aggregation kernel (sl_spmm_blockdiag_gather_f32).  ``materialize()`` is the plain gather for every
    other consumer."""

    def __init__(self, table: torch.Tensor, idx: torch.Tensor):
        _need_cuda(table, idx)
        assert table.dim() == 2 and table.dtype == torch.float32 and idx.dtype == torch.int32
        self.table, self.idx = table, idx.contiguous()
        self._dense = None

    @property
    def shape(self):
        return (int(self.idx.numel()), int(self.table.shape[1]))

    @property
    def device(self):
        return self.table.device

    @property
    def is_cuda(self):
        return True

    def materialize(self) -> torch.Tensor:
        if self._dense is None:
            self._dense = gather_rows(self.table, self.idx)
        return self._dense

    def gather_dropped(self, drop_p: float = 0.0):
        """dropout(table[idx]) in ONE pass, in rows padded with zeros to whole 128-byte lines: returns the [n, F] view
        of an [n, F_pad] buffer (F = 100 -> F_pad = 128; aligned rows for the SpMM tiles and the GEMM operand loads)
        and the dropout seed.  Falls back to gather + nn.functional.dropout for layouts the kernel does not take."""
        n, F = self.shape
        t = self.table
        if not (F % 4 == 0 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0):
            x = self.materialize()
            return (torch.nn.functional.dropout(x, drop_p, True) if drop_p > 0 else x), 0
        Fp = (F + 31) // 32 * 32
        buf = torch.empty(n, Fp, dtype=torch.float32, device=t.device)
        amax = torch.empty(n, dtype=torch.float32, device=t.device)
        seed = new_dropout_seed() if drop_p > 0 else 0
        with _timed(f"gather_F{F}", 8 * n * F + 4 * n, t.device):
            check(_lib.load().sl_gather_rows_drop_f32(t.data_ptr(), t.stride(0), self.idx.data_ptr(), n, F, float(drop_p), int(seed),
                                                      buf.data_ptr(), buf.stride(0), Fp, amax.data_ptr(), _stream(buf)))
        v = set_row_amax(buf[:, :F], amax)
        v._shd_pad_zero = Fp != F          # (the kernel writes zeros into the pad: the layer's products may read whole lines)
        return v, seed


def dense_rows(x):
    """The feature matrix itself, or the gathered rows of a LazyRows."""
    return x.materialize() if isinstance(x, LazyRows) else x


def can_fuse_gather(adj: "NormAdj", x) -> bool:
    """Layer 0 can read a LazyRows inside the block-diagonal SpMM: the adjacency must carry its block offsets and the
    rows must be float4-addressable."""
    c = adj.csr
    return (isinstance(x, LazyRows) and c.subg_off is not None and x._dense is None
            and x.shape[1] % 4 == 0 and x.shape[1] >= BLOCKDIAG_MIN_F and x.table.stride(0) % 4 == 0
            and x.table.stride(1) == 1 and x.table.data_ptr() % 16 == 0)


def spmm_gather(adj: "NormAdj", x: LazyRows, drop_p: float = 0.0, want_dense: bool = True):
    """(adj @ dropout(table[idx]), dropout(table[idx]) or None) in one kernel -- no autograd (layer-0 inputs carry no
    gradient; the fused layer nodes call this from their forward)."""
    c = adj.csr
    n, F = x.shape
    Y = torch.empty(n, F, dtype=torch.float32, device=x.device)
    Xo = torch.empty(n, F, dtype=torch.float32, device=x.device) if want_dense else None
    seed = new_dropout_seed() if drop_p > 0 else 0
    e = c.e
    nbytes = 4 * (n + 1) + 4 * e + (4 * e if adj.edge_w is not None else 0) + 4 * n + 4 * n * F * (2 + (1 if want_dense else 0))
    opt = lambda t: t.data_ptr() if t is not None else None
    with _timed(f"spmm_gather_F{F}", nbytes, x.device):
        check(_lib.load().sl_spmm_blockdiag_gather_f32(
            c.indptr.data_ptr(), c.indices.data_ptr(), opt(adj.edge_w), None, opt(adj.row_scale), opt(adj.col_scale),
            x.table.data_ptr(), x.table.stride(0), x.idx.data_ptr(), float(drop_p), int(seed), opt(Xo),
            Xo.stride(0) if Xo is not None else 0, Y.data_ptr(), Y.stride(0), n, F, c.subg_off.data_ptr(),
            c.subg_edge_off.data_ptr(), int(c.subg_off.numel()) - 1, c.max_subg_nodes, _stream(Y)))
    return Y, Xo, seed


def _ptr_array(ts: Sequence[Optional[torch.Tensor]]):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def can_fuse_out_dropout(F: int, seg: Optional[int] = None) -> bool:
    """Shapes whose act_norm runs on the vector kernels (the fused output dropout lives there):
    F a multiple of 4 up to 256, normalisation segments of a power-of-two number of float4 lanes."""
    seg = F if seg is None else seg
    if not (F % 4 == 0 and 16 <= F <= 256 and seg % 4 == 0 and F % seg == 0):
        return False
    lpr = 4                                  # lanes per row: the power of two covering F / 4 float4 lanes
    while lpr * 4 < F:
        lpr *= 2
    ls = lpr if seg == F else seg // 4       # lanes per normalisation segment
    # the instantiations of act_norm_launch (csrc/aggregate.hip)
    return (lpr, ls) in {(64, 64), (64, 32), (64, 16), (64, 8), (32, 32), (32, 16), (32, 8), (16, 16), (16, 8), (8, 8), (4, 4)}


def new_dropout_seed() -> int:
    """64-bit seed from torch's CPU generator (reproducible under torch.manual_seed, no device sync)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def _mix32(h: torch.Tensor) -> torch.Tensor:
    """murmur3 finaliser on int64 tensors holding 32-bit values."""
    M32 = 0xFFFFFFFF
    h = h ^ (h >> 16)
    h = _mul32(h, 0x85EBCA6B)
    h = h ^ (h >> 13)
    h = _mul32(h, 0xC2B2AE35)
    return (h ^ (h >> 16)) & M32


def _mul32(b: torch.Tensor, a: int) -> torch.Tensor:
    """(a * b) mod 2^32 without int64 overflow (a, b < 2^32)."""
    M32 = 0xFFFFFFFF
    return (((b & 0xFFFF) * a) + ((((b >> 16) * (a & 0xFFFF)) & 0xFFFF) << 16)) & M32


def dropout_keep_mask(n: int, F: int, p: float, seed: int, device) -> torch.Tensor:
    """The keep mask the kernels generate for (p, seed) -- the rule documented in include/shadow_hip.h (csrc/actnorm_common.h),
    restated in torch for the tests: one murmur finaliser per PAIR of columns, a 16-bit field per element."""
    M32 = 0xFFFFFFFF
    r = torch.arange(n, device=device, dtype=torch.int64).unsqueeze(1)
    c = torch.arange(F, device=device, dtype=torch.int64).unsqueeze(0)
    row = (_mix32((r & M32) ^ (seed & M32)) + (r >> 32) + ((seed >> 32) & M32)) & M32
    h = _mix32((row + _mul32(c >> 1, 0x9E3779B1)) & M32)
    field = torch.where((c & 1) == 1, h >> 16, h & 0xFFFF)
    # (the kernels take p as a C float: the threshold is that of the ROUNDED probability)
    import numpy as _np
    thr = int(min(max(float(_np.float32(p)) * 65536.0, 1.0), 65535.0))
    return field >= thr


def _is_dual(drop) -> bool:
    return len(drop) > 2 and bool(drop[2])


def _an_fwd(Zs, biases, codes, sc, of, seg, out_scale, drop=(0.0, 0)):
    """Returns out, or (out, out_dropped) when ``drop`` is (p, seed, True) -- the dual mode of sl_act_norm_fwd."""
    nb = len(Zs)
    n, F = Zs[0].shape
    out = torch.empty(n, F, dtype=torch.float32, device=Zs[0].device)
    out2 = torch.empty_like(out) if _is_dual(drop) else None
    # (tall outputs: the row maxima for the fp16 products of the layer that reads them, written in the same pass)
    amax = torch.empty(n, dtype=torch.float32, device=out.device) if (n >= AMAX_HANDOVER_ROWS and out.is_cuda) else None
    ld = (C.c_int64 * nb)(*[z.stride(0) for z in Zs])
    ac = (C.c_int * nb)(*codes)
    with _timed(f"act_norm_fwd_nb{nb}_F{F}", (nb + 1) * 4 * n * F, out.device):
        check(_lib.load().sl_act_norm_fwd(nb, _ptr_array(Zs), ld, _ptr_array(biases), ac, sc.data_ptr(), of.data_ptr(),
                                          n, F, seg, out_scale, out.data_ptr(), out.stride(0), float(drop[0]),
                                          int(drop[1]), out2.data_ptr() if out2 is not None else None,
                                          out2.stride(0) if out2 is not None else 0,
                                          amax.data_ptr() if amax is not None else None, _stream(out)))
    if amax is not None:
        set_row_amax(out2 if out2 is not None else out, amax)
    return out if out2 is None else (out, out2)


def _an_bwd(Zs, biases, codes, sc, of, seg, out_scale, douts, need_dz, want_dbias, drop=(0.0, 0), dz_out=None, row_idx=None,
            dz0_amax=None, dz_compact=False, t_out=None, amax_branch=0):
    """``douts``: the gradient(s) autograd handed over -- (dout,) or, in dual mode, (dout_plain, dout_dropped) with
    None for an output nothing consumed.  ``dz_out``: optional preallocated dZ destinations (column slices of a
    wider buffer are fine).  ``row_idx`` (int32 [m]): the gradients are compact [m, F] and belong to those rows (a read-out
    that took a few rows of the output, see RootsLink); the other rows of dZ are zero.  ``t_out`` = (branch, tensor [rows, F / seg]):
    also leave the per-segment dots dZ[branch] . Z[branch] there (sl_act_norm_bwd_rows_t: the GAT attention backward's t)."""
    nb = len(Zs)
    n, F = Zs[0].shape
    m = int(row_idx.numel()) if row_idx is not None else n
    dev = sc.device
    if not isinstance(douts, (tuple, list)):
        douts = (douts,)
    dout = _f32c(douts[0]) if douts[0] is not None else None
    dout2 = None
    if _is_dual(drop):
        dout2 = _f32c(douts[1]) if douts[1] is not None else None
        if dout2 is None:                   # only the plain output was used: no mask on the way back
            drop = (0.0, 0)
        elif dout2.stride(1) != 1:
            dout2 = dout2.contiguous()
    if dout is None and dout2 is None:
        dout = torch.zeros(m, F, dtype=torch.float32, device=dev)
    if dout is not None and dout.stride(1) != 1:
        dout = dout.contiguous()
    alloc = torch.zeros_like if row_idx is not None else torch.empty_like
    dZs = [((dz_out[i] if dz_out is not None and dz_out[i] is not None else alloc(z)) if nd else None)
           for i, (z, nd) in enumerate(zip(Zs, need_dz))]
    # (row_idx together with dz_out: the caller has cleared the rows outside row_idx itself, see sl_zero_slices;
    #  dz0_amax [n]: receives max |dZ_0[row]| of the rows written -- the caller clears it first when row_idx is given)
    dsc = torch.empty(nb, F, dtype=torch.float32, device=dev)
    dof = torch.empty(nb, F, dtype=torch.float32, device=dev)
    dbi = torch.empty(nb, F, dtype=torch.float32, device=dev) if want_dbias else None
    partial = torch.empty(2048 * nb * 3 * F, dtype=torch.float32, device=dev)
    ld = (C.c_int64 * nb)(*[z.stride(0) for z in Zs])
    ldd = (C.c_int64 * nb)(*[(d.stride(0) if d is not None else 0) for d in dZs])
    ac = (C.c_int * nb)(*codes)
    with _timed(f"act_norm_bwd_nb{nb}_F{F}" if row_idx is None else f"act_norm_bwd_rows_nb{nb}_F{F}", (2 * nb + 1) * 4 * m * F, dev):
        # (dz_compact: with row_idx, dZ / dz0_amax are [m, F] / [m] in the order of row_idx -- the caller's dz_out)
        check(_lib.load().sl_act_norm_bwd_rows_t(nb, _ptr_array(Zs), ld, _ptr_array(biases), ac, sc.data_ptr(), of.data_ptr(),
                                                 m, F, seg, out_scale, dout.data_ptr() if dout is not None else None,
                                                 dout.stride(0) if dout is not None else 0, _ptr_array(dZs), ldd,
                                                 dsc.data_ptr(), dof.data_ptr(), dbi.data_ptr() if dbi is not None else None,
                                                 partial.data_ptr(), float(drop[0]), int(drop[1]),
                                                 dout2.data_ptr() if dout2 is not None else None,
                                                 dout2.stride(0) if dout2 is not None else 0,
                                                 dz0_amax.data_ptr() if dz0_amax is not None else None,
                                                 row_idx.data_ptr() if row_idx is not None else None, 1 if dz_compact else 0,
                                                 t_out[0] if t_out is not None else -1, t_out[1].data_ptr() if t_out is not None else None,
                                                 int(amax_branch), _stream(Zs[0])))
    return dZs, dsc, dof, dbi


class _ActNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, offset, acts, seg, out_scale, drop, *Zs):
        Zs = [_f32c(z) for z in Zs]
        _need_cuda(scale, offset, *Zs)
        nb = len(Zs)
        n, F = Zs[0].shape
        sc = scale.reshape(nb, F).contiguous().float()
        of = offset.reshape(nb, F).contiguous().float()
        out = _an_fwd(Zs, [None] * nb, acts, sc, of, seg, out_scale, drop)
        ctx.save_for_backward(sc, of, *Zs)
        ctx.meta = (acts, seg, out_scale, scale.shape, offset.shape, drop)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, *dout):
        sc, of, *Zs = ctx.saved_tensors
        acts, seg, out_scale, sshape, oshape, drop = ctx.meta
        nb = len(Zs)
        need = ctx.needs_input_grad[6:6 + nb]
        dZs, dsc, dof, _ = _an_bwd(Zs, [None] * nb, acts, sc, of, seg, out_scale, dout, need, False, drop)
        return (dsc.reshape(sshape), dof.reshape(oshape), None, None, None, None, *dZs)


# split-bf16 MFMA GEMM (csrc/gemm.hip) for the tall feature x weight products; rocBLAS fp32 otherwise
# (SHADOW_GEMM_SPLIT_MIN_ROWS=1 sends every product the kernels can take through them -- the parity suite replays
#  the reference's small golden fixtures that way)
# (default 1024 rows: below that rocBLAS wins on GPU time; between 1 k and 8 k rows it is the HOST time of a rocBLAS
#  launch -- ~65 us of heuristics per call against ~10 us for the two ctypes calls -- that decides, measured on the
#  arxiv-shape GCN-3 / 32-root configuration: 2.42 -> 1.95 ms per step)
GEMM_SPLIT_MIN_ROWS = int(os.environ.get("SHADOW_GEMM_SPLIT_MIN_ROWS", "1024"))
GEMM_SPLIT = True


def mm_nt(A: torch.Tensor, B: torch.Tensor, min_rows: Optional[int] = None) -> torch.Tensor:
    """A[M,K] @ B[N,K]^T in fp32.  Tall products (M >= GEMM_SPLIT_MIN_ROWS, N <= 256) run on the bf16 matrix cores
    with the exact three-way operand split (fp32-level accuracy, see csrc/gemm.hip); the rest goes to
    rocBLAS.  ``min_rows``: the caller's own threshold (the row-sparse backward's small products: ROOT_GEMM_MIN_ROWS)."""
    M, K = A.shape
    N = B.shape[0]
    if not (GEMM_SPLIT and A.is_cuda and M >= (GEMM_SPLIT_MIN_ROWS if min_rows is None else min_rows) and N <= 256 and A.dtype == torch.float32
            and A.stride(1) == 1 and A.stride(0) % 4 == 0 and A.data_ptr() % 16 == 0):
        return A @ B.t()
    lib = _lib.load()
    Bc = B.detach()
    if Bc.stride(1) != 1:
        Bc = Bc.contiguous()
    packed = torch.empty(lib.sl_gemm_pack_bytes(N, K), dtype=torch.uint8, device=A.device)
    Cm = torch.empty(M, N, dtype=torch.float32, device=A.device)
    st = _stream(A)
    check(lib.sl_gemm_pack_b(Bc.data_ptr(), Bc.stride(0), N, K, packed.data_ptr(), st))
    # (keyed like the kernel instantiation rocprofv3 reports: one name per N and per "K % 32" variant)
    with _timed(f"gemm_nt_split_N{N}" + ("" if K % 32 == 0 else "_Ktail"), 4 * M * (K + N), A.device, flops=2 * M * K * N):
        check(lib.sl_gemm_nt_f32(A.data_ptr(), A.stride(0), packed.data_ptr(), Cm.data_ptr(), Cm.stride(0), M, N, K, st))
    return Cm


def weight_grad_f16(dZ: torch.Tensor, X: torch.Tensor, dz_amax: Optional[torch.Tensor] = None, x_amax: Optional[torch.Tensor] = None,
                    want_colsum: bool = False):
    """dW = dZ^T X (both 256 wide) on two fp16 pieces per element with the joint row scaling of sl_gemm_tn_f16; the row
    maxima come from the operands' producers (``get_row_amax``) or from one more pass (``row_amax``).
    ``want_colsum``: returns (dW, dZ.sum(0)) -- nn.Linear's bias gradient from the same pass over dZ."""
    n = dZ.shape[0]
    dz_amax = dz_amax if dz_amax is not None else (get_row_amax(dZ) if get_row_amax(dZ) is not None else row_amax(dZ))
    x_amax = x_amax if x_amax is not None else (get_row_amax(X) if get_row_amax(X) is not None else row_amax(X))
    lib = _lib.load()
    per = 256 * 256 + (256 if want_colsum else 0)
    partial = torch.empty(lib.sl_gemm_tn_slices(n) * per, dtype=torch.float32, device=dZ.device)
    dW = torch.empty(256, 256, dtype=torch.float32, device=dZ.device)
    cs = torch.empty(256, dtype=torch.float32, device=dZ.device) if want_colsum else None
    with _timed("gemm_tn_f16_N256", 4 * n * 512, dZ.device, flops=2 * n * 256 * 256):
        check(lib.sl_gemm_tn_f16(dZ.data_ptr(), dZ.stride(0), dz_amax.data_ptr(), X.data_ptr(), X.stride(0), x_amax.data_ptr(),
                                 dW.data_ptr(), n, 256, 256, partial.data_ptr(), cs.data_ptr() if cs is not None else None, _stream(dZ)))
    return (dW, cs) if want_colsum else dW


def weight_grad_f16_pair(dZ1: torch.Tensor, dZ2: torch.Tensor, X: torch.Tensor, dz_amax: torch.Tensor, x_amax: torch.Tensor,
                         want_colsum: bool = False):
    """(dZ1^T X, dZ2^T X) from one launch (sl_gemm_tn_f16_pair): dZ1, dZ2 share their pitch and the row maxima ``dz_amax``
    (two column blocks of one buffer, or two gradients with joint maxima); the two workgroups of a row slice share an XCD's L2,
    so X leaves HBM once.  ``want_colsum``: (dW1, dW2, dZ1.sum(0), dZ2.sum(0))."""
    n = dZ1.shape[0]
    assert dZ1.stride(0) == dZ2.stride(0)
    lib = _lib.load()
    per = 256 * 256 + (256 if want_colsum else 0)
    partial = torch.empty(2 * lib.sl_gemm_tn_slices(n) * per, dtype=torch.float32, device=X.device)
    dW1, dW2 = (torch.empty(256, 256, dtype=torch.float32, device=X.device) for _ in range(2))
    cs = [torch.empty(256, dtype=torch.float32, device=X.device) for _ in range(2)] if want_colsum else [None, None]
    opt = lambda t: t.data_ptr() if t is not None else None
    with _timed("gemm_tn_f16_pair_N256", 4 * n * 768, X.device, flops=2 * 2 * n * 256 * 256):
        check(lib.sl_gemm_tn_f16_pair(dZ1.data_ptr(), dZ2.data_ptr(), dZ1.stride(0), dz_amax.data_ptr(), X.data_ptr(), X.stride(0),
                                      x_amax.data_ptr(), dW1.data_ptr(), dW2.data_ptr(), n, 256, 256, partial.data_ptr(), opt(cs[0]), opt(cs[1]),
                                      _stream(X)))
    return (dW1, dW2, cs[0], cs[1]) if want_colsum else (dW1, dW2)


def weight_grad_f16_usable(dZ: torch.Tensor, X: torch.Tensor) -> bool:
    """Shapes sl_gemm_tn_f16 takes (the callers add: row maxima of both operands in hand)."""
    n = dZ.shape[0]
    return (GEMM_SPLIT and TN_F16 and dZ.is_cuda and dZ.shape[1] == 256 and X.shape[1] == 256 and n >= AMAX_HANDOVER_ROWS
            and dZ.dtype == torch.float32 and X.dtype == torch.float32 and dZ.stride(1) == 1 and X.stride(1) == 1
            and dZ.stride(0) % 4 == 0 and X.stride(0) % 4 == 0 and dZ.data_ptr() % 16 == 0 and X.data_ptr() % 16 == 0
            and -(-n // _lib.load().sl_gemm_tn_slices(n)) <= 3024)


TN_F16 = True


def weight_grad(dZ: torch.Tensor, X: torch.Tensor, want_colsum: bool = False, min_rows: Optional[int] = None):
    """dW = dZ^T X for tall inputs (K = number of batch nodes, hundreds of thousands).
    rocBLAS picks a 32-workgroup kernel for a plain 256 x n x 256 product; splitting n
    into a batched GEMM fills the chip (2x faster on MI355X) and the partial sums add
    in a fixed order.  ``want_colsum``: returns (dW, dZ.sum(0)) -- nn.Linear's bias gradient from the same pass over dZ."""
    if want_colsum:
        return _weight_grad_colsum(dZ, X, min_rows)
    n, Fo = dZ.shape
    Fi = X.shape[1]
    if (GEMM_SPLIT and dZ.is_cuda and n >= (GEMM_SPLIT_MIN_ROWS if min_rows is None else min_rows) and Fo <= 256 and Fi <= 256 and Fo % 4 == 0 and Fi % 4 == 0
            and dZ.dtype == torch.float32 and X.dtype == torch.float32 and dZ.stride(1) == 1 and X.stride(1) == 1
            and dZ.stride(0) % 4 == 0 and X.stride(0) % 4 == 0 and dZ.data_ptr() % 16 == 0 and X.data_ptr() % 16 == 0):
        lib = _lib.load()
        G = lib.sl_gemm_tn_slices(n)
        partial = torch.empty(G * Fo * Fi, dtype=torch.float32, device=dZ.device)
        dW = torch.empty(Fo, Fi, dtype=torch.float32, device=dZ.device)
        with _timed(f"gemm_tn_split_N{Fo}" + ("_K128" if Fi <= 128 else ""), 4 * n * (Fo + Fi), dZ.device, flops=2 * n * Fo * Fi):
            check(lib.sl_gemm_tn_f32(dZ.data_ptr(), dZ.stride(0), X.data_ptr(), X.stride(0), dW.data_ptr(), n, Fo, Fi,
                                     partial.data_ptr(), None, _stream(dZ)))
        return dW
    if n < 32768:
        return dZ.t() @ X
    dZ, X = dZ.contiguous(), X.contiguous()          # (column-slice views reach here from the fused SAGE node)
    S = min(256, max(2, n // 2048))
    c = n // S
    dW = torch.bmm(dZ[:S * c].view(S, c, Fo).transpose(1, 2), X[:S * c].view(S, c, Fi)).sum(0)
    if S * c < n:
        dW += dZ[S * c:].t() @ X[S * c:]
    return dW


def _tn_usable(dZ, X, min_rows=None) -> bool:
    n, Fo = dZ.shape
    Fi = X.shape[1]
    return (GEMM_SPLIT and dZ.is_cuda and n >= (GEMM_SPLIT_MIN_ROWS if min_rows is None else min_rows) and Fo <= 256 and Fi <= 256 and Fo % 4 == 0 and Fi % 4 == 0
            and dZ.dtype == torch.float32 and X.dtype == torch.float32 and dZ.stride(1) == 1 and X.stride(1) == 1
            and dZ.stride(0) % 4 == 0 and X.stride(0) % 4 == 0 and dZ.data_ptr() % 16 == 0 and X.data_ptr() % 16 == 0)


def _weight_grad_colsum(dZ, X, min_rows=None):
    if not _tn_usable(dZ, X, min_rows):
        return weight_grad(dZ, X, min_rows=min_rows), dZ.sum(0)
    n, Fo = dZ.shape
    Fi = X.shape[1]
    lib = _lib.load()
    G = lib.sl_gemm_tn_slices(n)
    partial = torch.empty(G * (Fo * Fi + Fo), dtype=torch.float32, device=dZ.device)
    dW = torch.empty(Fo, Fi, dtype=torch.float32, device=dZ.device)
    db = torch.empty(Fo, dtype=torch.float32, device=dZ.device)
    with _timed(f"gemm_tn_split_N{Fo}" + ("_K128" if Fi <= 128 else ""), 4 * n * (Fo + Fi), dZ.device, flops=2 * n * Fo * Fi):
        check(lib.sl_gemm_tn_f32(dZ.data_ptr(), dZ.stride(0), X.data_ptr(), X.stride(0), dW.data_ptr(), n, Fo, Fi,
                                 partial.data_ptr(), db.data_ptr(), _stream(dZ)))
    return dW, db


class PairLink:
    """Hand-over between a _LinearPair node (GAT's self / neighbour Linears of one input) and the node that consumes BOTH of
    its outputs and nothing else does (ops_gat._GatTail): when that node's backward ran row-sparse it leaves the two output
    gradients on the rows T here -- (rows32 [t], dza [t, N], dzb [t, N]) -- and hands autograd storage-less placeholders."""
    def __init__(self):
        self.filled = False
        self.rows32 = self.dza = self.dzb = self.dummy = self.levels = self.row_map = None

    def release(self):
        self.filled = False
        self.rows32 = self.dza = self.dzb = self.dummy = self.levels = self.row_map = None


class GatPre:
    """The GAT layer's attention terms that the paired Linear's kernel leaves with its outputs (sl_gemm_nt2_gat_f32): the
    node's SECOND output is then hn = act(z_neigh) instead of z_neigh, and u_s / u_n [n, heads] are here for ops_gat._GatTail --
    the only consumer such a pair may have (layers.GAT.forward asks for it only then)."""
    def __init__(self, attention, act_code: int, heads: int):
        self.attention, self.act_code, self.heads = attention, int(act_code), int(heads)
        self.u_s = self.u_n = None
        self.filled = False


# the GAT layer's per-node attention terms from the paired Linear's kernel (False: gat_node_fwd_kernel's pass; tests compare the two)
GAT_PAIR_TAIL = True


class _LinearPair(torch.autograd.Function):
    """Two nn.Linear of the SAME input -- GAT's self / neighbour transforms (shaDow/layers.py:604-611) -- as one autograd
    node on the fp16 two-piece kernels: forward = ONE two-product launch that reads X once per product and adds the
    biases as the tiles leave (sl_gemm_nt2_f32); backward: dX = [dZa | dZb] . [Wa ; Wb] as ONE K-concatenated product
    straight from the two gradient tensors (sl_gemm_nt_cat_f32: no dXa + dXb pass), the bias gradients from the
    weight-gradient kernel's pass over dZ (column sums of its A tiles)."""
    calls = 0                   # forward passes of the paired node (tests assert on it)
    gat_tail_calls = 0          # ... that also left hn / u_s / u_n

    @staticmethod
    def usable(X, Wa, Wb) -> bool:
        M, K = X.shape
        N = Wa.shape[0]
        return (GEMM_SPLIT and X.is_cuda and M >= GEMM_SPLIT_MIN_ROWS and Wa.shape == Wb.shape and X.dtype == torch.float32
                and Wa.dtype == torch.float32 and N % 32 == 0 and N <= 256 and K % 4 == 0 and 16 <= K <= 256
                and bool(_lib.load().sl_gemm_act_norm_supported(N, K)))

    @staticmethod
    def forward(ctx, X, Wa, ba, Wb, bb, pair=None, in_link=None, gat=None):
        """``pair`` (PairLink): the consumer of both outputs may leave their gradients on a few rows; ``in_link`` (RootsLink
        the producer of X published): the input gradient may then go down as (rows, values) too."""
        X = _f32c(X)
        if X.stride(0) % 4 or X.data_ptr() % 16:
            X = X.contiguous()
        ctx.pair, ctx.in_link = pair, (in_link if (in_link is not None and in_link.published) else None)
        lib = _lib.load()
        M, K = X.shape
        N = Wa.shape[0]
        dev, st = X.device, _stream(X)
        Ws = [w.detach() if w.stride(1) == 1 else w.detach().contiguous() for w in (Wa, Wb)]
        pack = torch.empty(2 * lib.sl_gemm_act_norm_pack_bytes(N, K), dtype=torch.uint8, device=dev)
        check(lib.sl_gemm_act_norm_pack(2, _ptr_array(Ws), (C.c_int64 * 2)(*[w.stride(0) for w in Ws]), N, K, pack.data_ptr(), None, 0, st))
        am = get_row_amax(X)
        if am is None and M >= AMAX_HANDOVER_ROWS:
            am = row_amax(X)
        Zs = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(2)]
        bs = [b.detach().contiguous() if b is not None else None for b in (ba, bb)]
        if gat is not None:
            # (z_self, hn = act(z_neigh)) and the attention's per-node terms from the same launch (GatPre)
            att = gat.attention.detach().float().contiguous()
            gat.u_s, gat.u_n = torch.empty(M, gat.heads, dtype=torch.float32, device=dev), torch.empty(M, gat.heads, dtype=torch.float32, device=dev)
            with _timed(f"gemm_nt2_gat_f16_N{N}" + ("" if K % 32 == 0 else "_Ktail"), 4 * M * (K + 2 * N + 2 * gat.heads), dev, flops=2 * 2 * M * K * N):
                check(lib.sl_gemm_nt2_gat_f32(X.data_ptr(), X.stride(0), am.data_ptr() if am is not None else None, pack.data_ptr(), M, N, K,
                                              _ptr_array(bs), Zs[0].data_ptr(), N, Zs[1].data_ptr(), N, att.data_ptr(), gat.act_code, gat.heads,
                                              gat.u_s.data_ptr(), gat.u_n.data_ptr(), st))
            gat.filled = True
            _LinearPair.gat_tail_calls += 1
        else:
          with _timed(f"gemm_nt2_f16_N{N}" + ("" if K % 32 == 0 else "_Ktail"), 4 * M * (K + 2 * N), dev, flops=2 * 2 * M * K * N):
            check(lib.sl_gemm_nt2_f32(2, _ptr_array([X, X]), (C.c_int64 * 2)(X.stride(0), X.stride(0)), _ptr_array([am, am]), pack.data_ptr(),
                                      M, N, K, _ptr_array(bs), _ptr_array(Zs), (C.c_int64 * 2)(N, N), st))
        _LinearPair.calls += 1
        ctx.save_for_backward(X, Wa, Wb)
        ctx.has_bias = (ba is not None, bb is not None)
        ctx.x_amax = am                  # (the weight gradients scale their fp16 pieces by it, sl_gemm_tn_f16)
        ctx.set_materialize_grads(False)
        return Zs[0], Zs[1]

    @staticmethod
    def _rows_backward(ctx, dZa, dZb):
        """Both output gradients are non-zero on the rows ``pair.rows32`` only (compact in ``pair.dza`` / ``pair.dzb``): weight and
        bias gradients from those rows, the input gradient on those rows -- handed down as (rows, values) when the producer of X
        published a RootsLink, scattered into a zero tensor otherwise."""
        X, Wa, Wb = ctx.saved_tensors
        pair = ctx.pair
        for g in (dZa, dZb):
            if g is None or g.data_ptr() != pair.dummy.data_ptr() or tuple(g.stride()) != (0, 0):
                raise RuntimeError("row-sparse GAT backward: an output of the paired Linear has a consumer besides the attention node")
        M, K = X.shape
        rows32, dza, dzb, levels, row_map = pair.rows32, pair.dza, pair.dzb, pair.levels, pair.row_map
        pair.release()
        Tl = rows32.long()
        XT = X.index_select(0, Tl)
        ng = ctx.needs_input_grad
        out = [None] * 8
        # (the library's own kernels from ROOT_GEMM_MIN_ROWS rows on; the bias gradient comes out of the weight gradient's pass over dZ)
        for i, dz in ((0, dza), (1, dzb)):
            want_w, want_b = ng[1 + 2 * i], ctx.has_bias[i] and ng[2 + 2 * i]
            if want_b:
                dW, db = weight_grad(dz, XT, want_colsum=True, min_rows=ROOT_GEMM_MIN_ROWS)
                out[1 + 2 * i], out[2 + 2 * i] = (dW if want_w else None), db
            elif want_w:
                out[1 + 2 * i] = weight_grad(dz, XT, min_rows=ROOT_GEMM_MIN_ROWS)
        if ng[0]:
            t, N = dza.shape
            lib = _lib.load()
            if (GEMM_SPLIT and t >= ROOT_GEMM_MIN_ROWS and K <= 256 and dza.stride(1) == 1 and dzb.stride(1) == 1 and dza.stride(0) % 4 == 0
                    and dzb.stride(0) % 4 == 0 and dza.data_ptr() % 16 == 0 and dzb.data_ptr() % 16 == 0 and lib.sl_gemm_act_norm_supported(K, 2 * N)):
                # [dZa | dZb] . [Wa ; Wb] on the rows T: the K-concatenated product of the dense pass (sl_gemm_nt_cat_f32)
                st = _stream(X)
                pack = torch.empty(lib.sl_gemm_act_norm_pack_bytes(K, 2 * N), dtype=torch.uint8, device=X.device)
                wa, wb = Wa.detach(), Wb.detach()
                check(lib.sl_gemm_act_norm_pack_b2(wa.data_ptr(), wa.stride(1), wa.stride(0), N, wb.data_ptr(), wb.stride(1), wb.stride(0),
                                                   K, 2 * N, pack.data_ptr(), st))
                dXT = torch.empty(t, K, dtype=torch.float32, device=X.device)
                check(lib.sl_gemm_nt_cat_f32(dza.data_ptr(), dza.stride(0), N, dzb.data_ptr(), dzb.stride(0), None, pack.data_ptr(), t, K, 2 * N,
                                             None, dXT.data_ptr(), dXT.stride(0), st))
            else:
                dXT = torch.addmm(dza @ Wa, dzb, Wb)           # [t, K]
            link = ctx.in_link
            if link is not None:
                link.rows32, link.grad, link.plan, link.levels = rows32, dXT, None, (levels or None)
                link.row_map = row_map
                link.dummy = placeholder(M, K, X.device)
                link.filled = True
                out[0] = link.dummy
            else:
                out[0] = torch.zeros(M, K, dtype=torch.float32, device=X.device).index_copy_(0, Tl, dXT)
        return tuple(out)

    @staticmethod
    def backward(ctx, dZa, dZb):
        if ctx.pair is not None and ctx.pair.filled:
            return _LinearPair._rows_backward(ctx, dZa, dZb)
        X, Wa, Wb = ctx.saved_tensors
        lib = _lib.load()
        M, K = X.shape
        N = Wa.shape[0]
        dev, st = X.device, _stream(X)
        # (row maxima of the K-concatenated operand: only when the kernel that wrote BOTH gradients left one joint array)
        ama, amb = (get_row_amax(d) if d is not None else None for d in (dZa, dZb))
        joint = ama if (ama is not None and amb is not None and ama.data_ptr() == amb.data_ptr()) else None
        dZs = [(_f32c(d).contiguous() if d is not None else torch.zeros(M, N, dtype=torch.float32, device=dev)) for d in (dZa, dZb)]
        ng = ctx.needs_input_grad
        dX = None
        if ng[0]:
            # image of [Wa^T | Wb^T]: "row" j = input feature j, K-concatenated over the two branches' outputs
            pack = torch.empty(lib.sl_gemm_act_norm_pack_bytes(K, 2 * N), dtype=torch.uint8, device=dev)
            wa, wb = Wa.detach(), Wb.detach()
            check(lib.sl_gemm_act_norm_pack_b2(wa.data_ptr(), wa.stride(1), wa.stride(0), N, wb.data_ptr(), wb.stride(1), wb.stride(0),
                                               K, 2 * N, pack.data_ptr(), st))
            dX = torch.empty(M, K, dtype=torch.float32, device=dev)
            with _timed(f"gemm_nt_f16_N{K}", 4 * M * (2 * N + K), dev, flops=2 * M * 2 * N * K):
                check(lib.sl_gemm_nt_cat_f32(dZs[0].data_ptr(), dZs[0].stride(0), N, dZs[1].data_ptr(), dZs[1].stride(0),
                                             joint.data_ptr() if joint is not None else None,
                                             pack.data_ptr(), M, K, 2 * N, None, dX.data_ptr(), dX.stride(0), st))
        out = [dX, None, None, None, None, None, None, None]
        # two fp16 pieces when the row maxima of both operands are in hand (the joint maxima bound either gradient's rows)
        f16 = joint is not None and ctx.x_amax is not None and weight_grad_f16_usable(dZs[0], X) and weight_grad_f16_usable(dZs[1], X)
        if f16 and ng[1] and ng[3] and dZs[0].stride(0) == dZs[1].stride(0):
            # both weight gradients (and both bias gradients) from one launch: the two products share X (sl_gemm_tn_f16_pair)
            want_b = [hb and ng[2 + 2 * i] for i, hb in enumerate(ctx.has_bias)]
            res = weight_grad_f16_pair(dZs[0], dZs[1], X, joint, ctx.x_amax, any(want_b))
            out[1], out[3] = res[0], res[1]
            if any(want_b):
                out[2], out[4] = (res[2] if want_b[0] else None), (res[3] if want_b[1] else None)
            return tuple(out)
        for i, (dz, hb) in enumerate(zip(dZs, ctx.has_bias)):
            want_w, want_b = ng[1 + 2 * i], hb and ng[2 + 2 * i]
            if want_b:
                dW, db = weight_grad_f16(dz, X, joint, ctx.x_amax, True) if f16 else weight_grad(dz, X, want_colsum=True)
                out[1 + 2 * i], out[2 + 2 * i] = (dW if want_w else None), db
            elif want_w:
                out[1 + 2 * i] = weight_grad_f16(dz, X, joint, ctx.x_amax) if f16 else weight_grad(dz, X)
        return tuple(out)


def linear_pair(X, lin_a: "torch.nn.Linear", lin_b: "torch.nn.Linear", gat: Optional[GatPre] = None):
    """(lin_a(X), lin_b(X)); one fused node when the shapes allow (see _LinearPair), two ``linear`` nodes otherwise.
    ``gat``: the caller is a GAT layer whose fused tail (ops_gat.gat_tail) is the ONLY consumer of the two outputs -- where the
    kernel takes the shape the second output is then act(lin_b(X)) and ``gat`` carries u_s / u_n (``gat.filled``)."""
    if isinstance(X, torch.Tensor) and X.dim() == 2 and _LinearPair.usable(X, lin_a.weight, lin_b.weight):
        # (row-sparse GAT backward: a PairLink travels with the two outputs, the producer's RootsLink with the input)
        pair = PairLink() if (SPARSE_TOP_BWD and ROOTS_SPARSE_GRAD) else None
        if gat is not None and not (GAT_PAIR_TAIL and _lib.load().sl_gemm_nt2_gat_supported(lin_a.weight.shape[0], gat.heads)):
            gat = None
        za, zb = _LinearPair.apply(X, lin_a.weight, lin_a.bias, lin_b.weight, lin_b.bias, pair, getattr(X, "_shadow_roots", None), gat)
        if pair is not None:
            za._shd_pair = zb._shd_pair = pair
        if gat is not None:
            zb._shd_gat_pre = gat
        return za, zb
    return linear(X, lin_a), linear(X, lin_b)


class _Linear(torch.autograd.Function):
    """nn.Linear with the split-K weight gradient."""
    @staticmethod
    def forward(ctx, X, W, b):
        X = _f32c(X).contiguous()
        ctx.save_for_backward(X, W)
        ctx.has_bias = b is not None
        Z = mm_nt(X, W)
        return Z + b if b is not None else Z

    @staticmethod
    def backward(ctx, dZ):
        X, W = ctx.saved_tensors
        dZ = _f32c(dZ).contiguous()
        dX = mm_nt(dZ, W.t()) if ctx.needs_input_grad[0] else None
        dW = weight_grad(dZ, X) if ctx.needs_input_grad[1] else None
        db = dZ.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dX, dW, db


def linear(X, lin: "torch.nn.Linear"):
    return _Linear.apply(X, lin.weight, lin.bias)


def gemm_act_norm_usable(Xs, Ws, seg, F) -> bool:
    """The GEMM-epilogue form of Linear + act + norm (sl_gemm_act_norm_fwd): tall fp32 operands of one shape, the
    normalisation over the whole row."""
    X0 = Xs[0]
    M, K = X0.shape
    if not (GEMM_SPLIT and X0.is_cuda and M >= GEMM_SPLIT_MIN_ROWS and seg == F and _lib.load().sl_gemm_act_norm_supported(F, K)):
        return False
    for x, w in zip(Xs, Ws):
        if not (x.shape == X0.shape and x.dtype == torch.float32 and x.stride(1) == 1 and x.stride(0) % 4 == 0
                and x.data_ptr() % 16 == 0 and tuple(w.shape) == (F, K) and w.dtype == torch.float32):
            return False
    return True


AMAX_HANDOVER_ROWS = 32768      # (kAmaxHandoverRows of csrc/layer_fused.hip)


def row_amax(A: torch.Tensor) -> torch.Tensor:
    """max_k |A[i, k]| per row: what the GEMM-epilogue kernels derive the fp16 operand scales from (sl_row_amax,
    include/shadow_hip.h).  Kernels that produce an operand leave it on the tensor (``set_row_amax``) instead."""
    n, K = A.shape
    am = torch.empty(n, dtype=torch.float32, device=A.device)
    check(_lib.load().sl_row_amax(A.data_ptr(), A.stride(0), n, K, am.data_ptr(), _stream(A)))    # (timed by the C entry itself)
    return am


def set_row_amax(t: torch.Tensor, amax: torch.Tensor) -> torch.Tensor:
    """Attach the row maxima a producer kernel wrote for ``t`` (valid until t is modified in place)."""
    t._shd_row_amax = (amax, t._version)
    return t


def get_row_amax(t: torch.Tensor) -> Optional[torch.Tensor]:
    a = getattr(t, "_shd_row_amax", None)
    if a is None or a[1] != t._version or a[0].shape[0] != t.shape[0] or a[0].device != t.device:
        return None
    return a[0]


def gemm_act_norm_fwd(Xs, Ws, biases, codes, sc, of, out_scale, drop):
    """Z_b = X_b W_b^T (kept for the backward pass) and out = out_scale * sum_b norm_b(act(Z_b + bias_b)) [+ the fused
    output dropout] from ONE kernel: the activation / normalisation runs in the GEMM's epilogue (csrc/gemm_fused.hip).
    Returns (Zs, out) or (Zs, (out, out_dropped)) in dual mode."""
    lib = _lib.load()
    nb = len(Xs)
    M, K = Xs[0].shape
    F = Ws[0].shape[0]
    dev = Xs[0].device
    st = _stream(Xs[0])
    pack = torch.empty(nb * lib.sl_gemm_act_norm_pack_bytes(F, K), dtype=torch.uint8, device=dev)
    wcs = [w.detach() if w.stride(1) == 1 else w.detach().contiguous() for w in Ws]
    check(lib.sl_gemm_act_norm_pack(nb, _ptr_array(wcs), (C.c_int64 * nb)(*[w.stride(0) for w in wcs]), F, K, pack.data_ptr(), None, 0, st))
    # (row maxima: left by the producer, one pass for a tall operand, or none -- the kernel then reads its rows twice)
    rsc = [get_row_amax(x) if get_row_amax(x) is not None else (row_amax(x) if M >= AMAX_HANDOVER_ROWS else None) for x in Xs]
    Zs = [torch.empty(M, F, dtype=torch.float32, device=dev) for _ in range(nb)]
    out = torch.empty(M, F, dtype=torch.float32, device=dev)
    out2 = torch.empty_like(out) if _is_dual(drop) else None
    lda = (C.c_int64 * nb)(*[x.stride(0) for x in Xs])
    ldz = (C.c_int64 * nb)(*[F] * nb)
    ac = (C.c_int * nb)(*codes)
    # algorithmic bytes: read every X_b, write every Z_b and the output(s); flops of the nb products
    nbytes = 4 * M * (nb * K + nb * F + F * (2 if out2 is not None else 1))
    with _timed(f"gemm_act_norm_fwd_nb{nb}_N{F}" + ("" if K % 32 == 0 else "_Ktail"), nbytes, dev, flops=2 * nb * M * K * F):
        check(lib.sl_gemm_act_norm_fwd(nb, _ptr_array(Xs), lda, _ptr_array(rsc), pack.data_ptr(), M, F, K, _ptr_array(Zs), ldz,
                                       _ptr_array(biases), ac, sc.data_ptr(), of.data_ptr(), float(out_scale), out.data_ptr(),
                                       out.stride(0), float(drop[0]), int(drop[1]), out2.data_ptr() if out2 is not None else None,
                                       out2.stride(0) if out2 is not None else 0, None, None, st))
    return Zs, (out if out2 is None else (out, out2))


def gemm_an_bwd(A, W, Zs, biases, codes, sc, of, drop=(0.0, 0), want_dbias=True, row_stats=None):
    """G = A @ W^T is the gradient of out = sum_b norm_b(act(Z_b + bias_b)) (through its fused output dropout when
    drop[0] > 0); returns (dZs, dscale, doffset, dbias) without ever writing G: the act_norm backward runs in the GEMM's
    epilogue (sl_gemm_an_bwd; what sl_sage_bwd_chain does between two GraphSAGE layers).  nb = 2 only."""
    lib = _lib.load()
    nb = len(Zs)
    M, K = A.shape
    N = W.shape[0]
    dev = A.device
    st = _stream(A)
    pack = torch.empty(lib.sl_gemm_act_norm_pack_bytes(N, K), dtype=torch.uint8, device=dev)
    Wc = W.detach().contiguous()
    check(lib.sl_gemm_act_norm_pack_b2(Wc.data_ptr(), Wc.stride(0), 1, K, Wc.data_ptr(), Wc.stride(0), 1, N, K, pack.data_ptr(), st))
    rsc = row_amax(A) if M >= AMAX_HANDOVER_ROWS else None
    dZs = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(nb)]
    dsc = torch.empty(nb, N, dtype=torch.float32, device=dev)
    dof = torch.empty(nb, N, dtype=torch.float32, device=dev)
    dbi = torch.empty(nb, N, dtype=torch.float32, device=dev) if want_dbias else None
    partial = torch.empty(lib.sl_gemm_an_bwd_partial_floats(M, N, nb), dtype=torch.float32, device=dev)
    ldz = (C.c_int64 * nb)(*[z.stride(0) for z in Zs])
    lddz = (C.c_int64 * nb)(*[N] * nb)
    ac = (C.c_int * nb)(*codes)
    nbytes = 4 * M * (K + 2 * nb * N)           # read A and every Z_b, write every dZ_b
    with _timed(f"gemm_an_bwd_nb{nb}_N{N}", nbytes, dev, flops=2 * M * K * N):
        check(lib.sl_gemm_an_bwd(A.data_ptr(), A.stride(0), rsc.data_ptr() if rsc is not None else None, pack.data_ptr(), M, N, K, nb, _ptr_array(Zs), ldz, _ptr_array(biases), ac,
                                 sc.data_ptr(), of.data_ptr(), 1.0, _ptr_array(dZs), lddz, dsc.data_ptr(), dof.data_ptr(),
                                 dbi.data_ptr() if dbi is not None else None, partial.data_ptr(), float(drop[0]), int(drop[1]), None,
                                 row_stats.data_ptr() if row_stats is not None else None, st))
    return dZs, dsc, dof, dbi


class ChainLink:
    """Hand-over between two consecutive GraphSAGE nodes whose only connection is lower.out -> upper.X (residue 'none' +
    centre pooling: nothing else reads the lower layer's output).  The upper node's input gradient IS the lower node's
    output gradient, so the upper node's K = 2F input-gradient GEMM runs the lower node's act_norm backward in its
    epilogue (sl_sage_bwd_chain) and leaves dZs / dZn / dscale / doffset / dbias here; what autograd carries between the
    two nodes is a storage-less placeholder (`dummy`)."""

    def __init__(self):
        self.published = False       # forward: the lower node has stored what the upper node's epilogue needs
        self.filled = False          # backward: the upper node has produced the lower node's dZ
        self.Zs = self.Zn = self.biases = self.sc = self.of = None
        self.act = 0
        self.drop = (0.0, 0)
        self.buf = self.dsc = self.dof = self.dbi = self.partial = self.dummy = self.amax = None
        self.stats = None            # [n, 4] (mean, 1 / std) per row and branch, left by the forward GEMM epilogue (or None)
        self.rows = None             # int32 row ids outside which the dZ in ``buf`` is zero (left by a row-sparse top pass), or None
        self.compact = None          # (dZs[T] [t, F], dZn[T] with a zero row behind [t + 1, F], plan): dZ on the rows T only, no ``buf``
        # dual-output lower layer (round 6): its PLAIN output feeds a read-out (ops.pool_and_roots), whose backward leaves the dense
        # gradient here -- autograd carries ``plain_dummy`` -- for the layer above's epilogue to add unmasked (sl_gemm_an_bwd_plain).
        # ``plain_index`` set: ``plain_grad`` is the [P + K, F] TABLE of a mean / sum pooling read-out's gradient and row i's gradient
        # is plain_grad[plain_index[i]] (POOL_GRAD_TABLE: the [n, F] expansion is never written)
        self.dual = False
        self.plain_grad = self.plain_dummy = self.plain_index = None
        # the LAST layer of a stack under a pooled read-out: no layer above, the link only carries the read-out's gradient table to
        # the layer's own act + norm backward (sl_act_norm_bwd_map)
        self.plain_only = False

    def publish(self, Zs, Zn, biases, sc, of, act, drop, stats=None):
        self.Zs, self.Zn, self.biases, self.sc, self.of, self.act, self.drop = Zs, Zn, biases, sc, of, int(act), drop
        self.stats = stats
        self.published = True

    def release(self):
        self.published = self.filled = False
        self.Zs = self.Zn = self.biases = self.sc = self.of = self.stats = self.rows = self.compact = None
        self.buf = self.dsc = self.dof = self.dbi = self.partial = self.dummy = self.amax = None
        self.plain_grad = self.plain_dummy = self.plain_index = None

    def plain_dense(self):
        """The plain output's gradient as an [n, F] tensor (a consumer that does not take the table form)."""
        if self.plain_index is None:
            return self.plain_grad
        return self.plain_grad.index_select(0, self.plain_index.long())


# The pooled read-out's gradient reaches a chained dual-output layer as a table [subgraphs + roots, F] and a row map (mean / sum
# pooling: every non-root row of a subgraph receives the same gradient row), not as the [n, F] tensor segment_pool_bwd writes:
# products-ppr-sage5 spends 59 us per layer writing that tensor and ~35 us reading it back.  False: the dense form.
POOL_GRAD_TABLE = True

# Dual-output GraphSAGE layers (a read-out that reads every layer: residue max / concat, mean / max pooling) chain their backward
# passes too: the lower layer's act + norm backward rides in the upper layer's input-gradient product, the plain output's gradient
# -- handed over by the read-out's pooling node -- added in the epilogue (round 6; False: every dual layer runs its own
# stand-alone act + norm backward, the round-5 path; tests compare the two)
CHAIN_DUAL = True


# Test tap: when a list, every fused Linear + act + norm node appends (pre-activations Z_b, biases) of its forward pass
# (tests/test_layers_gpu.py reads the relu sides of the run under test from it).  None in production.
Z_TAP = None


def _tap(Zs, biases):
    if Z_TAP is not None:
        # (the biases are parameters: copied, the optimiser updates them in place before the test reads the tap)
        Z_TAP.append(([z.detach() for z in Zs], [b.detach().clone() if b is not None else None for b in biases]))


class _LinearActNorm(torch.autograd.Function):
    """out = out_scale * sum_b norm_b(act_b(X_b W_b^T + bias_b)): the dense tail of a
    GCN / GraphSAGE / MLP layer as ONE autograd node.  GEMMs through mm_nt (split-bf16 MFMA kernel for
    the tall products, rocBLAS otherwise); bias add,
    activation, normalisation, branch sum and -- in backward -- dZ, dscale, doffset AND the
    bias gradients come from one HIP kernel pass each."""
    @staticmethod
    def forward(ctx, scale, offset, acts, seg, out_scale, nb, drop, *t):
        Xs = [_f32c(x).contiguous() for x in t[:nb]]
        Ws = list(t[nb:2 * nb])
        bs = list(t[2 * nb:3 * nb])
        _need_cuda(scale, offset, *Xs, *Ws)
        F = Ws[0].shape[0]
        sc = scale.reshape(nb, F).contiguous().float()
        of = offset.reshape(nb, F).contiguous().float()
        bsc = [b.detach().contiguous() if b is not None else None for b in bs]
        if gemm_act_norm_usable(Xs, Ws, seg, F):
            Zs, out = gemm_act_norm_fwd(Xs, Ws, bsc, acts, sc, of, out_scale, drop)
        else:
            Zs = [mm_nt(x, w) for x, w in zip(Xs, Ws)]
            out = _an_fwd(Zs, bsc, acts, sc, of, seg, out_scale, drop)
        _tap(Zs, bsc)
        ctx.save_for_backward(sc, of, *Xs, *Ws, *Zs, *[b if b is not None else sc.new_empty(0) for b in bsc])
        ctx.meta = (acts, seg, out_scale, nb, scale.shape, offset.shape, [b is not None for b in bs], drop)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, *dout):
        acts, seg, out_scale, nb, sshape, oshape, has_b, drop = ctx.meta
        sv = ctx.saved_tensors
        sc, of = sv[0], sv[1]
        Xs, Ws, Zs, bs = sv[2:2 + nb], sv[2 + nb:2 + 2 * nb], sv[2 + 2 * nb:2 + 3 * nb], sv[2 + 3 * nb:2 + 4 * nb]
        biases = [b if hb else None for b, hb in zip(bs, has_b)]
        dZs, dsc, dof, dbi = _an_bwd(list(Zs), biases, acts, sc, of, seg, out_scale, dout, [True] * nb, any(has_b), drop)
        ng = ctx.needs_input_grad
        dXs = [mm_nt(dz, w.t()) if ng[7 + i] else None for i, (dz, w) in enumerate(zip(dZs, Ws))]
        dWs = [weight_grad(dz, x) if ng[7 + nb + i] else None for i, (dz, x) in enumerate(zip(dZs, Xs))]
        dbs = [dbi[i] if (has_b[i] and ng[7 + 2 * nb + i]) else None for i in range(nb)]
        return (dsc.reshape(sshape), dof.reshape(oshape), None, None, None, None, None, *dXs, *dWs, *dbs)


class RootsLink:
    """Hand-over between the LAST GraphSAGE node of a stack and a read-out that takes only a few rows of its output
    (centre pooling, residue 'none': shaDow/layers.py:159-163 reads feat[idx_targets]).  The reference's ``feat[rows]`` hands
    autograd a zero-filled [n, F] gradient; here ``select_roots`` leaves (rows, their gradient) on the link, autograd carries
    a storage-less placeholder, and the node's backward runs its act_norm backward on those rows only (every other row of
    dZ is zero and is cleared, not computed) -- sl_sage_bwd_chain(d_dout_rows)."""
    def __init__(self):
        self.published = self.filled = False
        self.rows32 = self.grad = self.dummy = None
        self.csr = None              # the batch CSR of the publishing node (a TopBackwardPlan is built from it when none came along)
        self.plan = None             # tail.TopBackwardPlan of the selected rows: the node's backward may then run row-sparse
        self.want_levels = False     # the publishing node works from tail.build_backward_levels instead (GAT)
        self.levels = None           # remaining levels of a row-sparse backward pass, the one for THIS node's rows first
        self.row_map = None          # int32 [n] or None: position of every batch row in rows32, -1 = absent (RectLevel.in_map32)

    def release(self):
        self.filled = False
        self.rows32 = self.grad = self.dummy = self.plan = self.levels = self.row_map = None


ROOTS_SPARSE_GRAD = True
# The top GraphSAGE layer's backward pass on the rows its gradient is non-zero on (tail.TopBackwardPlan): exact, see there.
# Off: the dense kernels stream the 99.6 %-zero gradient (SHADOW_SPARSE_TOP_BWD=0; bench.py reports that step time beside `value`).
SPARSE_TOP_BWD = os.environ.get("SHADOW_SPARSE_TOP_BWD", "1") != "0"
# below: ~20 small launches cost more host time than the three dense kernels cost GPU time (products shape, 128 roots = 36 k rows:
# 2.14 ms / step dense, 2.87 with the row-sparse pass; 1 024 roots = 289 k rows: 7.28 -> 6.59)
SPARSE_TOP_BWD_MIN_ROWS = int(os.environ.get("SHADOW_SPARSE_TOP_BWD_MIN_ROWS", "131072"))
# the row-sparse pass from the whole-stack node (_SageStack._sparse_top: the layers below it in ONE C call) instead of the
# layer-by-layer nodes (False: tests compare the two)
SPARSE_TOP_STACK = True
# the row-sparse passes' small products on the library's own kernels from this many rows on, torch.mm / rocBLAS below.  Same-box
# A/B (scripts/ab_root_gemm.sh, scripts/ab_top_stack.sh): the GAT stack's products over the rows T (~20 k rows) 11.63 -> 11.45 ms
# per step on the own kernels (and the bias gradient comes out of the weight gradient's pass); the GraphSAGE top layer's four
# products over the 1 024 roots 6.37 -> 6.46 (a workgroup covers 128 rows: 8 workgroups); host time equal either way
# (scripts/host_breakdown.py with HB_ROOT_GEMM).
ROOT_GEMM_MIN_ROWS = 2048
# a level of tail.build_backward_levels is kept while its input set is at most this share of the batch (built on the spot by
# select_roots when the batch brings none)
BACKWARD_LEVELS_FRAC = 0.25


class _SelectRoots(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, rows, link, grad_on=True):
        ctx.link, ctx.n, ctx.F = link, int(f.shape[0]), int(f.shape[1])
        ctx.save_for_backward(rows)
        ctx.set_materialize_grads(False)
        # the row sets of the row-sparse top-layer backward: handed over with the rows (built by the minibatch extractor on its
        # prefetch stream) or, for hand-made batches, built here (two host syncs)
        ctx.plan = ctx.levels = None
        # (autograd is switched off INSIDE a Function's forward: the caller's grad mode comes in as an argument.  Evaluation /
        #  a frozen producer: no backward pass will ask for the row sets -- no plan, no host syncs)
        want = bool(SPARSE_TOP_BWD and grad_on and ctx.needs_input_grad[0])
        if want and link is not None and link.published and ctx.n >= SPARSE_TOP_BWD_MIN_ROWS and link.csr is not None and link.want_levels:
            lv = getattr(rows, "_shd_bwd_levels", None)
            if lv is None:
                from . import tail
                lv = tail.build_backward_levels(link.csr, rows, frac=BACKWARD_LEVELS_FRAC)
            ctx.levels = lv if (lv and lv[0].r == int(rows.numel())) else None
        elif want and link is not None and link.published and ctx.n >= SPARSE_TOP_BWD_MIN_ROWS and link.csr is not None:
            plan = getattr(rows, "_shd_top_plan", None)
            if plan is None or not plan.matches(link.csr, int(rows.numel())):
                from . import tail
                plan = tail.TopBackwardPlan(link.csr, rows)
            ctx.plan = plan if plan.matches(link.csr, int(rows.numel())) else None       # (not ok: a multigraph root row -- dense pass)
        return f.index_select(0, rows)

    @staticmethod
    def backward(ctx, dsel):
        (rows,) = ctx.saved_tensors
        link = ctx.link
        if dsel is None:
            dsel = torch.zeros(rows.numel(), ctx.F, dtype=torch.float32, device=rows.device)
        if link is not None and link.published:
            link.rows32 = rows.to(torch.int32)
            link.grad = _f32c(dsel).contiguous()
            link.plan = ctx.plan
            link.levels = list(ctx.levels) if ctx.levels else None
            link.dummy = placeholder(ctx.n, ctx.F, dsel.device)
            link.filled = True
            return link.dummy, None, None, None
        dense = torch.zeros(ctx.n, ctx.F, dtype=dsel.dtype, device=dsel.device)
        dense.index_add_(0, rows, dsel)
        return dense, None, None, None


def select_roots(f: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
    """f[rows] (the read-out's row select).  When ``f`` came from a node that published a RootsLink the gradient travels
    as (rows, values) instead of a zero-filled [n, F] tensor.  ``rows`` must be distinct (the roots of a collated batch
    are: every subgraph owns its rows) -- the sparse backward WRITES the selected rows of dZ, it does not accumulate."""
    link = getattr(f, "_shadow_roots", None)
    if link is None or not link.published or not ROOTS_SPARSE_GRAD:
        return f[rows]
    return _SelectRoots.apply(f, rows, link, torch.is_grad_enabled())


def _at_dzn_on_rows(adj: "NormAdj", plan, dZnT: torch.Tensor, n: int, Fo: int):
    """(A^T dZn [n, Fo], its row maxima [n]) for a gradient dZn that lives on the rows T of ``plan`` (``dZnT`` [t + 1, Fo]: the
    compact rows and a zero row behind them).  Transposed structure: row scale <- the adjacency's column scale and vice versa."""
    lib = _lib.load()
    c = adj.csr
    dev = dZnT.device
    f32 = dict(dtype=torch.float32, device=dev)
    st = _stream(dZnT)
    opt = lambda t_: t_.data_ptr() if t_ is not None else None
    t = plan.t
    AtdZn = rows_empty(0, n, Fo, dev)
    ew = adj.edge_w
    filtered = plan.f_indptr is not None and 128 < Fo <= 256
    amx = torch.empty(n, **f32) if filtered else torch.zeros(n, **f32)      # (the filtered pass writes every row's maximum)
    if filtered:
        # the transposed structure filtered to the columns T (tail.TopBackwardPlan): the rows' kept entries in their
        # original order -- the sums of the row-mapped walk below without its zero terms -- gathered from the compact
        # gradient (22 MB: L2 / Infinity Cache), row maxima from the same pass
        ef = plan.f_nnz
        rsT = adj.row_scale.index_select(0, plan.T32.long()) if adj.row_scale is not None else None
        nbytes = 4 * (n + 1) + 4 * ef + (8 * ef if ew is not None else 0) + 4 * n * Fo + 4 * t * Fo
        with _timed(f"spmm_rows_F{Fo}", nbytes, dev):
            check(lib.sl_spmm_csr_amax_f32(plan.f_indptr.data_ptr(), plan.f_indices.data_ptr(), opt(ew),
                                           plan.f_perm.data_ptr() if ew is not None else None, opt(adj.col_scale), opt(rsT),
                                           dZnT.data_ptr(), dZnT.stride(0), AtdZn.data_ptr(), Fo, n, Fo, amx.data_ptr(), st))
        _SageDense.filtered_spmm_calls += 1
    else:
      nbytes = 4 * (n + 1) + 4 * c.e + (4 * c.e if ew is not None else 0) + 4 * n * Fo + 4 * t * Fo
      with _timed(f"spmm_rows_F{Fo}", nbytes, dev):
        check(lib.sl_spmm_blockdiag_rows_f32(c.transposed[0].data_ptr(), c.transposed[1].data_ptr(), opt(ew), c.transposed[2].data_ptr() if ew is not None else None,
                                             opt(adj.col_scale), opt(adj.row_scale), dZnT.data_ptr(), dZnT.stride(0), plan.rowmap.data_ptr(),
                                             AtdZn.data_ptr(), Fo, n, Fo, c.spmm_blocks[0].data_ptr(), c.spmm_blocks[1].data_ptr(), int(c.spmm_blocks[0].numel()) - 1, c.spmm_blocks[2],
                                             amx.data_ptr(), t, st))
    return AtdZn, amx


class _SageDense(torch.autograd.Function):
    """The whole dense part of a GraphSAGE layer (shaDow/layers.py:471-483) as ONE autograd node:
        out = norm_0(act(X Ws^T + bs)) + norm_1(act((A X) Wn^T + bn))
    so that the backward pass can use  dX = [dZs | A^T dZn] . [Ws ; Wn]  -- one GEMM with K = 2F that writes
    dX once -- instead of two GEMMs, a transposed SpMM on the product and an add.  Consecutive nodes can be CHAINED
    (``link_down`` / ``link_up``, see ChainLink): dX is then never written at all."""
    fused_calls = 0          # one-call entries taken (tests assert on it: "the path that is timed is the path that is tested")
    chained_calls = 0        # backward passes that produced the lower layer's dZ in the GEMM epilogue

    @staticmethod
    def forward(ctx, X, adj, Ws, bs, Wn, bn, scale, offset, acts, drop, lazy=None, in_drop=0.0, link_down=None, link_up=None,
                link_roots=None):
        """``lazy`` (a LazyRows; X is then a dummy): layer 0 -- the aggregation kernel gathers the features, applies the
        layer's input dropout ``in_drop`` and leaves the dense copy the self Linear and the weight gradients read."""
        _need_cuda(Ws, Wn, scale, offset)
        c = adj.csr
        F = Ws.shape[0]
        one_call = False
        if lazy is not None:
            AX, X, _seed = spmm_gather(adj, lazy, drop_p=in_drop, want_dense=True)
        else:
            X = _f32c(X)                   # (may be the [n, F] view of line-padded rows: every consumer takes a row pitch)
            _need_cuda(X)
            one_call = _SageDense._fusable(X, Ws, Wn)
            if one_call:
                AX = None                  # the one-call entry below computes it
            else:
                AX = _spmm_raw(c.indptr, c.indices, adj.edge_w, None, adj.row_scale, adj.col_scale, X, c.n,
                               c.spmm_blocks, row_entries_bound=c.row_entries_bound)
        sc = scale.reshape(2, F).contiguous().float()
        of = offset.reshape(2, F).contiguous().float()
        bsc = [b.detach().contiguous() if b is not None else None for b in (bs, bn)]
        ctx.x_amax = None
        ctx_stats = None
        if AX is None:
            ctx.x_amax = get_row_amax(X)        # (the backward's weight gradients scale their fp16 pieces by it, sl_gemm_tn_f16)
            # (row statistics for the chained backward of this layer: only when a layer above will chain into it)
            want_stats = (link_up is not None and not link_up.plain_only and CHAIN_SAGE_BWD and (not _is_dual(drop) or (CHAIN_DUAL and F > 128)) and ROW_STATS_HANDOVER
                          and _lib.load().sl_gemm_act_norm_supported(F, X.shape[1]) and F % 32 == 0)
            AX, Zs, Zn, out, ctx_stats = _SageDense._fused_forward(X, adj, Ws, Wn, bsc, sc, of, acts, drop, want_stats)
            _SageDense.fused_calls += 1
        elif gemm_act_norm_usable([X, AX], [Ws, Wn], F, F):
            (Zs, Zn), out = gemm_act_norm_fwd([X, AX], [Ws, Wn], bsc, acts, sc, of, 1.0, drop)
        else:
            Zs, Zn = mm_nt(X, Ws), mm_nt(AX, Wn)
            out = _an_fwd([Zs, Zn], bsc, acts, sc, of, F, 1.0, drop)
        _tap([Zs, Zn], bsc)
        ctx.save_for_backward(X, AX, Ws, Wn, Zs, Zn, sc, of, *[b if b is not None else sc.new_empty(0) for b in bsc])
        ctx.adj = adj
        ctx.meta = (acts, drop, scale.shape, offset.shape, [b is not None for b in (bs, bn)], one_call)
        # chaining (both ends need the one-call entries and a non-dual output)
        ctx.link_down = link_down if (link_down is not None and link_down.published and one_call and CHAIN_SAGE_BWD) else None
        ctx.link_roots = None
        if link_roots is not None and one_call and not _is_dual(drop) and F % 4 == 0 and 16 <= F <= 256:
            link_roots.published = True
            link_roots.csr = c
            ctx.link_roots = link_roots
        ctx.link_up = None
        # (published only when THIS node's backward will take the one-call entry -- the only consumer of the dZ the layer
        # above leaves on the link: a layer-0 input wider than 256 (Flickr 500, Yelp 300) or frozen weights run kernel by
        # kernel, and the layer above must then write a real dX)
        # (a dual-output layer publishes too -- CHAIN_DUAL, the 256-wide epilogue instantiation: the read-out's pooling node will
        #  leave the plain output's gradient on the link)
        if (link_up is not None and one_call and CHAIN_SAGE_BWD and (not _is_dual(drop) or (CHAIN_DUAL and 128 < F <= 256 and F % 32 == 0))
                and F % 4 == 0 and 16 <= F <= 256
                and _SageDense._bwd_fusable(ctx.needs_input_grad, one_call, X.shape[1], F, AX)):
            link_up.publish(Zs, Zn, bsc, sc, of, acts[0], drop, ctx_stats)
            link_up.dual = _is_dual(drop)
            ctx.link_up = link_up
        ctx.set_materialize_grads(False)
        fire_deferred()
        return out

    @staticmethod
    def _fusable(X, Ws, Wn):
        Fo, Fi = Ws.shape
        return (FUSED_LAYER_CALLS and GEMM_SPLIT and X.shape[0] >= max(1, GEMM_SPLIT_MIN_ROWS)
                and Fo % 4 == 0 and Fo <= 256
                and Fi % 4 == 0 and X.dtype == torch.float32 and X.stride(1) == 1 and X.stride(0) % 4 == 0 and X.data_ptr() % 16 == 0
                and Ws.stride(1) == 1 and Wn.stride(1) == 1 and Ws.dtype == torch.float32 and Wn.shape == Ws.shape)

    @staticmethod
    def _bwd_fusable(ng, one_call, Fi, Fo, AX):
        """Whether the backward pass takes the one-call entry (sl_sage_bwd_chain): decided from what the forward knows."""
        return bool(one_call and ng[2] and ng[4] and Fi <= 256 and AX.stride(0) % 4 == 0 and AX.data_ptr() % 16 == 0
                    and (not ng[0] or Fo % 32 == 0))

    @staticmethod
    def _fused_forward(X, adj, Ws, Wn, biases, sc, of, acts, drop, want_stats=False):
        lib = _lib.load()
        n, Fi = X.shape
        Fo = Ws.shape[0]
        dev = X.device
        pitch = X.stride(0) if (X.stride(0) != Fi and X.stride(0) % 32 == 0) else Fi
        AX = torch.empty(n, pitch, dtype=torch.float32, device=dev)[:, :Fi]
        Zs = torch.empty(n, Fo, dtype=torch.float32, device=dev)
        Zn = torch.empty(n, Fo, dtype=torch.float32, device=dev)
        out = torch.empty(n, Fo, dtype=torch.float32, device=dev)
        out2 = torch.empty(n, Fo, dtype=torch.float32, device=dev) if _is_dual(drop) else None
        pack = torch.empty(lib.sl_sage_pack_bytes(n, Fi, Fo), dtype=torch.uint8, device=dev)
        x_amax = get_row_amax(X)                                   # left by the kernel that produced X (gather / the layer below)
        out_amax = torch.empty(n, dtype=torch.float32, device=dev)
        a = _adj_struct(adj, False)
        opt = lambda t: t.data_ptr() if t is not None else None
        # (the GEMM-epilogue kernel writes them: same eligibility as inside sl_sage_fwd)
        stats = (torch.empty(n, 4, dtype=torch.float32, device=dev)
                 if (want_stats and X.data_ptr() % 16 == 0 and X.stride(0) % 4 == 0 and AX.data_ptr() % 16 == 0 and AX.stride(0) % 4 == 0) else None)
        check(lib.sl_sage_fwd(C.byref(a), X.data_ptr(), X.stride(0), Fi, Fo, Ws.data_ptr(), Ws.stride(0), opt(biases[0]),
                              Wn.data_ptr(), Wn.stride(0), opt(biases[1]), sc.data_ptr(), of.data_ptr(), int(acts[0]), float(drop[0]),
                              int(drop[1]), AX.data_ptr(), AX.stride(0), Zs.data_ptr(), Zn.data_ptr(), out.data_ptr(), opt(out2),
                              opt(x_amax), out_amax.data_ptr(), pack.data_ptr(), 1 if getattr(X, "_shd_pad_zero", False) else 0,
                              opt(stats), _stream(X)))
        set_row_amax(out if out2 is None else out2, out_amax)      # (of the tensor the next layer's GEMM reads)
        return AX, Zs, Zn, (out if out2 is None else (out, out2)), stats

    @staticmethod
    def _fused_backward(ctx, dout, X, AX, Ws, Wn, Zs, Zn, sc, of, biases, acts, drop, has_b, want_dx):
        """One C call for the whole backward pass.  With ``ctx.link_up`` filled the act_norm backward is skipped (the layer
        above produced dZ); with ``ctx.link_down`` the input gradient is not written: the epilogue of its GEMM produces
        the dZ of the layer below instead (returns a storage-less placeholder for dX)."""
        lib = _lib.load()
        n, Fo = Zs.shape
        Fi = X.shape[1]
        dev = Zs.device
        f32 = dict(dtype=torch.float32, device=dev)
        up, down = ctx.link_up, ctx.link_down
        dz_ready = up is not None and up.filled
        lr0 = ctx.link_roots
        if (SPARSE_TOP_BWD and not dz_ready and lr0 is not None and lr0.filled and lr0.plan is not None and down is not None and want_dx
                and down.Zs.shape == (n, Fi) and Fo % 32 == 0 and Fi % 4 == 0 and float(drop[0]) == 0.0 and n >= SPARSE_TOP_BWD_MIN_ROWS
                and lr0.plan.matches(ctx.adj.csr, int(lr0.rows32.numel()))):
            g = dout[0] if isinstance(dout, (tuple, list)) else dout
            if g is None or g.data_ptr() != lr0.dummy.data_ptr() or tuple(g.stride()) != (0, 0):
                raise RuntimeError("sparse read-out gradient: the layer's output has a consumer besides the read-out "
                                   "(its gradient is not the placeholder); set ops.ROOTS_SPARSE_GRAD = False")
            return _SageDense._sparse_top_backward(ctx, lr0, down, X, AX, Ws, Wn, Zs, Zn, sc, of, biases, acts, has_b)
        d0 = d1 = dout_rows = dout_map = None
        if dz_ready:
            if up.dual:
                # (dual-output layer: the dropped output's gradient must be the chain's placeholder, the plain output's the one
                #  the read-out's pooling node left -- both already inside the dZ the layer above produced)
                gp, g = dout[0], dout[1]
                if gp is not None and (up.plain_dummy is None or gp.data_ptr() != up.plain_dummy.data_ptr() or tuple(gp.stride()) != (0, 0)):
                    raise RuntimeError("chained GraphSAGE backward: the lower layer's plain output has a consumer that did not hand its "
                                       "gradient to the chain; set ops.CHAIN_DUAL = False")
            else:
                g = dout[0] if isinstance(dout, (tuple, list)) else dout
            if g is None or g.data_ptr() != up.dummy.data_ptr() or tuple(g.stride()) != (0, 0):
                raise RuntimeError("chained GraphSAGE backward: the lower layer's output has a consumer besides the layer above "
                                   "(its gradient is not the chain's placeholder); build the model with chaining off")
            if up.compact is not None:
                # the layer above ran its row-sparse pass: this layer's dZs / dZn exist on the rows T only
                res = _SageDense._compact_dz_backward(ctx, up, down, X, AX, Ws, Wn, want_dx)
                if res is not None:
                    return res
                _SageDense._densify(up, n, Fo)                 # (no chained layer below / shapes the row-mapped kernels do not take)
            buf, dsc, dof, dbi = up.buf, up.dsc, up.dof, up.dbi
            an_partial = None
        else:
            douts = dout if isinstance(dout, (tuple, list)) else (dout,)
            lr = ctx.link_roots
            if lr is not None and lr.filled:
                # the read-out handed over (rows, gradient): autograd carried a storage-less placeholder
                g = douts[0]
                if g is None or g.data_ptr() != lr.dummy.data_ptr() or tuple(g.stride()) != (0, 0):
                    raise RuntimeError("sparse read-out gradient: the layer's output has a consumer besides the read-out "
                                       "(its gradient is not the placeholder); set ops.ROOTS_SPARSE_GRAD = False")
                dout_rows, douts = lr.rows32, (lr.grad,)
            if (up is not None and (up.dual or up.plain_only) and up.plain_grad is not None and douts[0] is not None and up.plain_dummy is not None
                    and douts[0].data_ptr() == up.plain_dummy.data_ptr()):
                # the pooling node left the plain gradient on the link but the layer above did not chain: it is consumed here
                # (a gradient TABLE stays one when this layer's own act + norm backward can read through the row map)
                if (up.plain_index is not None and dout_rows is None and lib.sl_act_norm_vector_layout(Fo, Fo)
                        and up.plain_grad.stride(0) == Fo and up.plain_grad.data_ptr() % 16 == 0):
                    dout_map = up.plain_index
                    douts = (up.plain_grad,) + tuple(douts[1:])
                else:
                    douts = (up.plain_dense(),) + tuple(douts[1:])
            d0 = _f32c(douts[0]).contiguous() if douts[0] is not None else None
            if _is_dual(drop):
                d1 = _f32c(douts[1]).contiguous() if douts[1] is not None else None
                if d1 is None:
                    drop = (0.0, 0)
            if d0 is None and d1 is None:
                d0, dout_rows = torch.zeros(n, Fo, **f32), None
            dbi = torch.empty(2, Fo, **f32) if any(has_b) else None
            dsc, dof = torch.empty(2, Fo, **f32), torch.empty(2, Fo, **f32)
            buf = torch.empty(n, 3 * Fo, **f32)
            an_partial = torch.empty(2048 * 2 * 3 * Fo, **f32)
        chain = down is not None and want_dx and down.Zs.shape == (n, Fi) and Fo % 32 == 0
        if chain and down.dual:
            # (the plain output's gradient must be on the link by now -- the read-out's nodes run before any conv layer's -- and dense)
            pg, pi = down.plain_grad, down.plain_index
            chain = bool(CHAIN_DUAL and pg is not None and pg.dim() == 2 and pg.shape[1] == Fi and pg.stride(1) == 1 and pg.stride(0) == Fi
                         and pg.data_ptr() % 16 == 0 and 128 < Fi <= 256
                         and (pg.shape[0] == n if pi is None else (pi.numel() == n and pi.dtype == torch.int32 and pi.is_contiguous())))
        below = None
        if chain:
            down.buf = torch.empty(n, 3 * Fi, **f32)
            down.dsc, down.dof = torch.empty(2, Fi, **f32), torch.empty(2, Fi, **f32)
            down.dbi = torch.empty(2, Fi, **f32) if any(b is not None for b in down.biases) else None
            down.partial = torch.empty(lib.sl_sage_chain_partial_floats(n, Fi), **f32)
            down.amax = torch.empty(n, **f32)
            opt = lambda t: t.data_ptr() if t is not None else None
            below = _lib.SlSageBelow(down.Zs.data_ptr(), down.Zn.data_ptr(), opt(down.biases[0]), opt(down.biases[1]),
                                     down.sc.data_ptr(), down.of.data_ptr(), down.act, float(down.drop[0]), int(down.drop[1]), Fi,
                                     down.buf.data_ptr(), down.dsc.data_ptr(), down.dof.data_ptr(), opt(down.dbi),
                                     down.partial.data_ptr(), down.amax.data_ptr(), opt(down.stats),
                                     down.plain_grad.data_ptr() if down.dual else None,
                                     down.plain_index.data_ptr() if (down.dual and down.plain_index is not None) else None, Fi)
        dX = torch.empty(n, Fi, **f32) if (want_dx and not chain) else None
        # dZ non-zero on a few rows only (the layer above ran its row-sparse pass): dWs = dZs[T]^T X[T], dWn = dZn[T]^T (A X)[T] on
        # those rows (21 k of 289 k) instead of the paired kernel over all of them
        sparse_rows = up.rows if (dz_ready and up.rows is not None and SPARSE_TOP_BWD) else None
        if sparse_rows is not None:
            Tl = sparse_rows.long()
            dWs = weight_grad(buf[:, :Fo].index_select(0, Tl), X.index_select(0, Tl))
            dWn = weight_grad(buf[:, 2 * Fo:].index_select(0, Tl), AX.index_select(0, Tl))
            tn_partial = None
        else:
            dWs, dWn = torch.empty(Fo, Fi, **f32), torch.empty(Fo, Fi, **f32)
            tn_partial = torch.empty((2 if ctx.x_amax is not None else 1) * lib.sl_gemm_tn_slices(n) * Fo * Fi, **f32)   # (both weight gradients in one launch)
        pack = torch.empty(lib.sl_sage_pack_bytes(n, Fi, Fo), dtype=torch.uint8, device=dev)
        a = _adj_struct(ctx.adj, want_dx)
        opt = lambda t: t.data_ptr() if t is not None else None
        check(lib.sl_sage_bwd_chain(C.byref(a), X.data_ptr(), X.stride(0), AX.data_ptr(), AX.stride(0), Zs.data_ptr(), Zn.data_ptr(), Fi,
                                    Fo, Ws.data_ptr(), Ws.stride(0), opt(biases[0]), Wn.data_ptr(), Wn.stride(0), opt(biases[1]),
                                    sc.data_ptr(), of.data_ptr(), int(acts[0]), float(drop[0]), int(drop[1]), opt(d0), opt(d1), opt(dX),
                                    dWs.data_ptr() if sparse_rows is None else None, dWn.data_ptr() if sparse_rows is None else None,
                                    opt(dbi), opt(dsc), opt(dof), buf.data_ptr(), opt(an_partial),
                                    opt(tn_partial), pack.data_ptr(), 1 if dz_ready else 0,
                                    C.byref(below) if below is not None else None,
                                    up.amax.data_ptr() if (dz_ready and up.amax is not None) else None,
                                    dout_rows.data_ptr() if dout_rows is not None else None,
                                    int(dout_rows.numel()) if dout_rows is not None else 0, opt(ctx.x_amax), opt(dout_map), _stream(Zs)))
        if dz_ready:
            up.release()
        if dout_rows is not None:
            ctx.link_roots.release()
        if chain:
            down.dummy = placeholder(n, Fi, dev)       # what autograd hands to the node below: no storage behind it
            down.filled = True
            dX = down.dummy
            _SageDense.chained_calls += 1
        return dX, dWs, dWn, dbi, dsc, dof

    filtered_spmm_calls = 0  # ... whose transposed aggregate ran over the structure filtered to T
    sparse_top_calls = 0     # backward passes of a top layer that ran on the rows R u N(R) only
    compact_dz_calls = 0     # backward passes of the layer below such a pass that took dZ on the rows T only

    @staticmethod
    def _densify(up, n, F):
        """Full-height [dZs | . | dZn] from the compact rows a row-sparse top pass left (the general fallback)."""
        dZsT, dZnT, plan = up.compact
        f32 = dict(dtype=torch.float32, device=dZsT.device)
        up.buf = torch.empty(n, 3 * F, **f32)
        check(_lib.load().sl_zero_slices(up.buf.data_ptr(), up.buf.data_ptr() + 8 * F, 3 * F, n, F, _stream(dZsT)))
        Tl = plan.T32.long()
        up.buf[:, :F].index_copy_(0, Tl, dZsT)
        up.buf[:, 2 * F:].index_copy_(0, Tl, dZnT[:plan.t])
        up.amax = None                  # (the C entry takes one more pass over dZs for its row maxima)
        up.rows, up.compact = plan.T32, None

    @staticmethod
    def _compact_dz_backward(ctx, up, down, X, AX, Ws, Wn, want_dx):
        """Backward pass of the layer BELOW a row-sparse top pass: dZs / dZn are given on the rows T (``up.compact``), zero elsewhere.
            dWs = dZs[T]^T X[T],  dWn = dZn[T]^T (A X)[T]                                   (21 k of 289 k rows)
            dX  = [dZs | A^T dZn] . [Ws ; Wn] = (A^T dZn) Wn  +  scatter_T(dZs[T] Ws)
        -- the dense half as a K = F product over A^T dZn (transposed block-diagonal SpMM whose staging reads dZn through a row
        map: the rows outside T share one zero row), the other half on the rows T, added in the GEMM's epilogue before it runs
        the act + norm backward of the layer below (sl_gemm_an_bwd_corr).  No [n, F] tensor of zeros is written or read.
        None: not applicable (the caller densifies)."""
        lib = _lib.load()
        dZsT, dZnT, plan = up.compact
        n, Fo = ctx.saved_tensors[4].shape
        Fi = X.shape[1]
        c = ctx.adj.csr
        off, eoff, mn = c.spmm_blocks
        chain = down is not None and want_dx and down.Zs.shape == (n, Fi) and Fo % 32 == 0
        if not (chain and off is not None and Fo >= BLOCKDIAG_MIN_F and Fo % 4 == 0 and lib.sl_gemm_act_norm_supported(Fi, Fo)
                and plan.rowmap.numel() == n):
            return None
        dev = X.device
        f32 = dict(dtype=torch.float32, device=dev)
        st = _stream(X)
        opt = lambda t_: t_.data_ptr() if t_ is not None else None
        adj = ctx.adj
        t = plan.t
        Tl = plan.T32.long()
        dWs = weight_grad(dZsT, X.index_select(0, Tl))
        dWn = weight_grad(dZnT[:t], AX.index_select(0, Tl))
        AtdZn, amx = _at_dzn_on_rows(adj, plan, dZnT, n, Fo)
        corr = mm_nt(dZsT, Ws.t())                                  # dZs[T] Ws  [t, Fi]
        pack = torch.empty(lib.sl_gemm_act_norm_pack_bytes(Fi, Fo), dtype=torch.uint8, device=dev)
        check(lib.sl_gemm_act_norm_pack_b2(Wn.data_ptr(), 1, Wn.stride(0), Fo, Wn.data_ptr(), 1, Wn.stride(0), Fi, Fo, pack.data_ptr(), st))
        down.buf = torch.empty(n, 3 * Fi, **f32)
        down.dsc, down.dof = torch.empty(2, Fi, **f32), torch.empty(2, Fi, **f32)
        down.dbi = torch.empty(2, Fi, **f32) if any(b is not None for b in down.biases) else None
        partial = torch.empty(lib.sl_gemm_an_bwd_partial_floats(n, Fi, 2), **f32)
        down.amax = torch.empty(n, **f32)
        ld2 = (C.c_int64 * 2)(Fi, Fi)
        ld3 = (C.c_int64 * 2)(3 * Fi, 3 * Fi)
        ac = (C.c_int * 2)(down.act, down.act)
        dZb = (C.c_void_p * 2)(down.buf.data_ptr(), down.buf.data_ptr() + 8 * Fi)
        nbytes = 4 * n * (Fo + 4 * Fi) + 4 * t * Fi                 # read A^T dZn and both Z of the layer below, write its two dZ (+ the addend)
        with _timed(f"gemm_an_bwd_corr_nb2_N{Fi}", nbytes, dev, flops=2 * n * Fo * Fi):
            check(lib.sl_gemm_an_bwd_corr(AtdZn.data_ptr(), Fo, amx.data_ptr(), pack.data_ptr(), n, Fi, Fo, 2, _ptr_array([down.Zs, down.Zn]), ld2,
                                          _ptr_array(down.biases), ac, down.sc.data_ptr(), down.of.data_ptr(), 1.0, dZb, ld3,
                                          down.dsc.data_ptr(), down.dof.data_ptr(), opt(down.dbi), partial.data_ptr(), float(down.drop[0]),
                                          int(down.drop[1]), down.amax.data_ptr(), opt(down.stats), corr.data_ptr(), corr.stride(0),
                                          plan.rowmap.data_ptr(), t, st))
        down.partial = partial
        down.dummy = placeholder(n, Fi, dev)
        down.filled = True
        dsc, dof, dbi = up.dsc, up.dof, up.dbi
        up.release()
        _SageDense.chained_calls += 1
        _SageDense.compact_dz_calls += 1
        return down.dummy, dWs, dWn, dbi, dsc, dof

    @staticmethod
    def _sparse_top_backward(ctx, lr, down, X, AX, Ws, Wn, Zs, Zn, sc, of, biases, acts, has_b):
        """The top layer of a GraphSAGE stack under a row-selecting read-out: its output gradient lives on the roots R
        (``lr.grad`` [P, Fo]), so dZs / dZn are zero outside R, dWs = dZs[R]^T X[R], dWn = dZn[R]^T (A X)[R], and the input
        gradient dX = dZs Ws + A^T (dZn Wn) is zero outside T = R u N(R) (tail.TopBackwardPlan).  The layer BELOW is chained
        (ChainLink): its act_norm backward runs on the rows T of dX -- through its fused output dropout mask -- and leaves
        dZs / dZn (zero elsewhere), the parameter gradients and the row maxima where sl_sage_bwd_chain's epilogue would.
        Replaces, on ~7 % of the rows, the transposed SpMM + K = 2F GEMM-epilogue + weight-gradient kernels that otherwise
        stream the zeros (0.99 ms of the 7.4 ms products step); same gradients (tests/test_layers_gpu.py::
        test_sparse_top_layer_backward_equals_dense)."""
        lib = _lib.load()
        n, Fo = Zs.shape
        Fi = X.shape[1]
        dev = Zs.device
        f32 = dict(dtype=torch.float32, device=dev)
        plan = lr.plan
        R = plan.rows64
        (dZsR, dZnR), dsc, dof, dbi = _an_bwd([Zs.index_select(0, R), Zn.index_select(0, R)], biases, acts, sc, of, Fo, 1.0, (lr.grad,),
                                              [True, True], any(has_b), (0.0, 0))
        dWs = weight_grad(dZsR, X.index_select(0, R), min_rows=ROOT_GEMM_MIN_ROWS)
        dWn = weight_grad(dZnR, AX.index_select(0, R), min_rows=ROOT_GEMM_MIN_ROWS)
        # dX on the rows T: a row gets its root's neighbour term through the edge (root, row), the root itself the self term
        GS = (mm_nt(dZnR, Wn.t(), min_rows=ROOT_GEMM_MIN_ROWS), mm_nt(dZsR, Ws.t(), min_rows=ROOT_GEMM_MIN_ROWS))
        adj = ctx.adj
        opt = lambda t_: t_.data_ptr() if t_ is not None else None
        dXT = torch.empty(plan.t, Fi, **f32)
        check(lib.sl_top_dx(GS[0].data_ptr(), GS[1].data_ptr(), Fi, plan.T32.data_ptr(), plan.slot.data_ptr(), plan.epos.data_ptr(),
                            plan.self_idx.data_ptr(), plan.targets32.data_ptr(), opt(adj.edge_w), opt(adj.row_scale), opt(adj.col_scale),
                            plan.t, Fi, dXT.data_ptr(), Fi, _stream(Zs)))
        # the layer below: act_norm backward on the rows T of its output gradient; its dZs / dZn stay COMPACT ([t, F], a zero
        # row behind dZn for the row-mapped transposed SpMM): nothing of height n is cleared or written here
        dZsT = torch.empty(plan.t, Fi, **f32)
        dZnT = torch.empty(plan.t + 1, Fi, **f32)
        dZnT[plan.t].zero_()
        _dz, down.dsc, down.dof, down.dbi = _an_bwd([down.Zs, down.Zn], down.biases, (down.act, down.act), down.sc, down.of, Fi, 1.0,
                                                   (dXT,), [True, True], any(b is not None for b in down.biases), down.drop,
                                                   dz_out=[dZsT, dZnT[:plan.t]], row_idx=plan.T32, dz_compact=True)
        down.compact = (dZsT, dZnT, plan)
        down.buf = down.amax = None
        down.partial = None
        down.rows = plan.T32           # (dZs / dZn of the layer below are zero outside T: its weight gradients need those rows only)
        lr.release()
        down.dummy = placeholder(n, Fi, dev)
        down.filled = True
        _SageDense.chained_calls += 1
        _SageDense.sparse_top_calls += 1
        return down.dummy, dWs, dWn, dbi, dsc, dof

    @staticmethod
    def backward(ctx, *dout):
        X, AX, Ws, Wn, Zs, Zn, sc, of, b0, b1 = ctx.saved_tensors
        acts, drop, sshape, oshape, has_b, one_call = ctx.meta
        adj = ctx.adj
        n, F = Zs.shape
        ng = ctx.needs_input_grad
        biases = [b if hb else None for b, hb in zip((b0, b1), has_b)]
        Fi = X.shape[1]
        # (the forward's decision, not a re-evaluation: a KernelTimer entered between the passes must not mix the paths)
        fused_bwd = _SageDense._bwd_fusable(ng, one_call, Fi, F, AX)
        lr = ctx.link_roots
        if lr is not None and lr.filled and not fused_bwd:
            # the read-out left (rows, gradient) on the link but this pass runs kernel by kernel: the dense form after all
            g = dout[0]
            if g is None or g.data_ptr() != lr.dummy.data_ptr() or tuple(g.stride()) != (0, 0):
                raise RuntimeError("sparse read-out gradient: the layer's output has a consumer besides the read-out")
            dense = torch.zeros(n, F, dtype=torch.float32, device=Zs.device)
            dense.index_add_(0, lr.rows32.long(), lr.grad)
            dout = (dense,) + tuple(dout[1:])
            lr.release()
        if fused_bwd:
            dX, dWs, dWn, dbi, dsc, dof = _SageDense._fused_backward(ctx, dout, X, AX, Ws, Wn, Zs, Zn, sc, of, biases, acts, drop,
                                                                     has_b, bool(ng[0]))
            dbs = dbi[0] if (has_b[0] and ng[3] and dbi is not None) else None
            dbn = dbi[1] if (has_b[1] and ng[5] and dbi is not None) else None
            return (dX, None, dWs, dbs, dWn, dbn, dsc.reshape(sshape), dof.reshape(oshape), None, None, None, None, None, None, None)
        if ctx.link_up is not None and ctx.link_up.filled:
            raise RuntimeError("chained GraphSAGE backward: the layer above filled this layer's dZ but the one-call path is off")
        # dZs lands in the left half of one [n, 2F] buffer; A^T dZn goes into the right half
        buf = torch.empty(n, 2 * F, dtype=torch.float32, device=Zs.device) if ng[0] else None
        dz_out = [buf[:, :F], None] if buf is not None else None
        (dZs, dZn), dsc, dof, dbi = _an_bwd([Zs, Zn], biases, acts, sc, of, F, 1.0, dout, [True, True], any(has_b),
                                            drop, dz_out)
        dX = None
        if ng[0]:
            c = adj.csr
            ti, tx, tp = c.transposed
            _spmm_raw(ti, tx, adj.edge_w, tp if adj.edge_w is not None else None, adj.col_scale, adj.row_scale, dZn,
                      c.n, c.spmm_blocks, out=buf[:, F:], row_entries_bound=c.row_entries_bound)
            dX = mm_nt(buf, torch.cat([Ws.t(), Wn.t()], dim=1))
        dWs = weight_grad(dZs, X) if ng[2] else None
        dWn = weight_grad(dZn, AX) if ng[4] else None
        dbs = dbi[0] if (has_b[0] and ng[3]) else None
        dbn = dbi[1] if (has_b[1] and ng[5]) else None
        return dX, None, dWs, dbs, dWn, dbn, dsc.reshape(sshape), dof.reshape(oshape), None, None, None, None, None, None, None


# One C call per direction for a whole stack of chained GraphSAGE layers (sl_sage_stack_fwd / sl_sage_stack_bwd) instead of one
# per layer: the same per-layer entries in the same order (bit-identical), minus ~100 us of Python, ctypes and autograd work per
# layer and direction -- which is what bounds a step at the reference's own batch sizes.  SHADOW_SAGE_STACK=0: layer by layer.
SAGE_STACK = os.environ.get("SHADOW_SAGE_STACK", "1") != "0"


class _SlotList:
    """[n, F] tensors addressed like the rows of a [k, n, F] tensor (shape / device / indexing): the whole-stack node's saved
    forward products as separate allocations (STACK_SEPARATE_SLOTS)."""
    def __init__(self, ts):
        self.ts = ts
        self.shape = (len(ts),) + tuple(ts[0].shape)
        self.device = ts[0].device

    def __getitem__(self, k):
        return self.ts[k]


# The whole-stack node's saved forward products as one allocation per tensor (True) or as the slots of ONE [4 L - 1, n, F] tensor.
# One tensor is 5.6 GB at 289 k rows, and the batches' row counts differ by a few per cent: whenever a batch exceeded every batch
# before it the caching allocator had to hipMalloc a new multi-GB block -- on some boxes that cost the HOST 2 - 3 ms per step
# averaged over a 60-step run (same box, scripts/ab_stack_slots.sh: 9.03 / 8.38 ms per step against 6.34 / 6.33 with separate
# 296 MB allocations and 6.35 / 6.31 for the layer-by-layer nodes; on other boxes 6.18 / 6.21 against 6.14 / 6.17).
STACK_SEPARATE_SLOTS = os.environ.get("SHADOW_STACK_SEPARATE_SLOTS", "1") != "0"
ROW_QUANTUM = 16384


def rows_empty(lead: int, n: int, width: int, device) -> torch.Tensor:
    """A float32 tensor [lead, n, width] (lead = 0: [n, width]) carved from the front of an allocation sized for n rounded up to
    ROW_QUANTUM rows: consecutive batches (whose n differ by a few per cent) request the SAME number of bytes, so the caching
    allocator hands the same block back instead of growing by a new multi-hundred-MB block at every new maximum."""
    k = max(1, lead)
    # (the quantum follows the batch: 1/16 of the next power of two up to ROW_QUANTUM -- a 1 100-row batch rounds up to 1 152
    #  rows, not to 16 384)
    q = min(ROW_QUANTUM, max(64, (1 << max(n - 1, 1).bit_length()) // 16))
    cap = -(-max(n, 1) // q) * q
    flat = torch.empty(k * cap * width, dtype=torch.float32, device=device)
    t = flat[:k * n * width]
    return t.view(k, n, width) if lead else t.view(n, width)




class _SageStack(torch.autograd.Function):
    """The conv loop of DeepGNN.forward (shaDow/models.py:193-197) over L GraphSAGE layers (layers.py:471-483) plus the
    read-out's row select (layers.py:159-163) as ONE autograd node: forward = sl_sage_stack_fwd, backward = sl_sage_stack_bwd,
    i.e. the chained one-call layer passes of _SageDense issued from C.  Preconditions (``stack_usable``): residue 'none' +
    centre pooling on a node task (nothing but layer l + 1 reads layer l's output, nothing but the row select reads the last
    one), one hidden width F with F % 32 == 0 <= 256, no dual output."""
    calls = 0
    sparse_top_calls = 0     # backward passes whose two top layers ran row-sparse from here (round 5)

    @staticmethod
    def forward(ctx, X0, adj, rows, meta, grad_on, plan, *params):
        lib = _lib.load()
        L = len(meta)
        n, F0 = X0.shape
        F = params[0].shape[0]
        dev = X0.device
        f32 = dict(dtype=torch.float32, device=dev)
        pitch0 = X0.stride(0) if (X0.stride(0) != F0 and X0.stride(0) % 32 == 0) else F0
        AX0 = torch.empty(n, pitch0, **f32)
        ctx.plan, ctx.meta_l = plan, meta
        keep_all = bool(grad_on) and any(ctx.needs_input_grad)   # (no backward pass will come -- no_grad keeps needs_input_grad True for live parameters --: nothing is kept: one Zs / Zn / A X slot, two `out` slots in turn)
        # per layer Zs, Zn, out; then A X of the layers 1 .. L - 1
        if keep_all and STACK_SEPARATE_SLOTS:
            big = _SlotList([rows_empty(0, n, F, dev) for _ in range(4 * L - 1)])        # (one allocation per saved tensor, as the layer-by-layer nodes make them)
        else:
            big = torch.empty(4 * L - 1, n, F, **f32) if keep_all else torch.empty(5 if L > 1 else 3, n, F, **f32)
        slot_z = (lambda l: 3 * l) if keep_all else (lambda l: 0)
        slot_out = (lambda l: 3 * l + 2) if keep_all else (lambda l: 2 + (l & 1) if L > 1 else 2)
        slot_ax = (lambda l: 3 * L + l - 1) if keep_all else (lambda l: 4)
        amax = torch.empty(L, n, **f32)
        # (row statistics for the chained backward: the layers another layer chains into, as _SageDense.forward decides)
        stats_ok = CHAIN_SAGE_BWD and ROW_STATS_HANDOVER and bool(lib.sl_gemm_act_norm_supported(F, F))
        stats0_ok = CHAIN_SAGE_BWD and ROW_STATS_HANDOVER and bool(lib.sl_gemm_act_norm_supported(F, F0)) and AX0.stride(0) % 4 == 0
        stats = torch.empty(max(1, L - 1), n, 4, **f32) if (keep_all and L > 1 and (stats_ok or stats0_ok)) else None
        arr = (_lib.SlSageStackLayer * L)()
        sp = (lambda k: big[k].data_ptr()) if isinstance(big, _SlotList) else (lambda k, base=big.data_ptr(), step=n * F * 4: base + k * step)
        for l in range(L):
            Ws, bs, Wn, bn, sc, of = params[6 * l:6 * l + 6]
            y = arr[l]
            y.Ws, y.Wn, y.scale, y.offset = Ws.data_ptr(), Wn.data_ptr(), sc.data_ptr(), of.data_ptr()
            y.bs = bs.data_ptr() if bs is not None else None
            y.bn = bn.data_ptr() if bn is not None else None
            y.ldws, y.ldwn = Ws.stride(0), Wn.stride(0)
            y.Fin, y.Fout = (F0 if l == 0 else F), F
            y.act, y.drop_p, y.drop_seed = meta[l]
            if l == 0:
                y.AX, y.ldax = AX0.data_ptr(), AX0.stride(0)
            else:
                y.AX, y.ldax = sp(slot_ax(l)), F
            y.Zs, y.Zn, y.out = sp(slot_z(l)), sp(slot_z(l) + 1), sp(slot_out(l))
            y.out_amax = amax.data_ptr() + l * n * 4
            if stats is not None and l < L - 1 and (stats_ok if l else stats0_ok):
                y.row_stats = stats.data_ptr() + l * n * 16
        x0_amax = get_row_amax(X0)
        pack = torch.empty(lib.sl_sage_stack_pack_bytes(n, L, arr), dtype=torch.uint8, device=dev)
        a = _adj_struct(adj, False)
        check(lib.sl_sage_stack_fwd(C.byref(a), X0.data_ptr(), X0.stride(0), x0_amax.data_ptr() if x0_amax is not None else None,
                                    1 if getattr(X0, "_shd_pad_zero", False) else 0, L, arr, pack.data_ptr(), _stream(X0)))
        if Z_TAP is not None and keep_all:
            for l in range(L):
                _tap([big[3 * l], big[3 * l + 1]], [params[6 * l + 1], params[6 * l + 3]])
        ctx.save_for_backward(X0, *[p for p in params if p is not None])
        ctx.has = [p is not None for p in params]
        ctx.rows = rows
        ctx.adj, ctx.arr, ctx.L, ctx.x0_amax = adj, arr, L, x0_amax
        ctx.keep = (AX0, big, amax, stats)                   # (the forward products the descriptors point into)
        ctx.set_materialize_grads(False)
        _SageStack.calls += 1
        _SageStack.last_slots = int(big.shape[0])             # ([n, F] slots this call allocated: tests hold the evaluation footprint)
        _SageDense.fused_calls += L                          # (the one-call layer entries ran L times, from C)
        fire_deferred()
        out = big[slot_out(L - 1)]
        return out.index_select(0, rows) if rows is not None else out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        saved = ctx.saved_tensors
        X0 = saved[0]
        it = iter(saved[1:])
        params = [next(it) if h else None for h in ctx.has]
        L, arr, adj, rows = ctx.L, ctx.arr, ctx.adj, ctx.rows
        if arr is None:
            raise RuntimeError("the layer-stack node was already back-propagated through (its forward products are released after the "
                               "first backward pass; retain_graph is not supported on this path: set SHADOW_SAGE_STACK=0)")
        n, F0 = X0.shape
        F = params[0].shape[0]
        dev = X0.device
        f32 = dict(dtype=torch.float32, device=dev)
        r = int(rows.numel()) if rows is not None else n
        d = _f32c(dout).contiguous() if dout is not None else torch.zeros(r, F, **f32)
        want_dx0 = bool(ctx.needs_input_grad[0])
        dX0 = torch.empty(n, F0, **f32) if want_dx0 else None
        dW = torch.empty(2 * max(1, L - 1), F, F, **f32)     # dWs, dWn of the layers 1 .. L - 1
        dW0 = torch.empty(2, F, F0, **f32)
        ds = torch.empty(2 * L, 2, F, **f32)                 # per layer dscale, doffset [2, F]
        db = torch.empty(2 * L, F, **f32)                    # per layer dbias of the self / neighbour Linear
        buf = rows_empty(2, n, 3 * F, dev)
        am = torch.empty(2, n, **f32)
        an_partial = torch.empty(2048 * 2 * 3 * F, **f32)
        chain_partial = torch.empty(max(lib.sl_sage_chain_partial_floats(n, F), 1), **f32) if L > 1 else None
        sl = lib.sl_gemm_tn_slices(n)
        tn_partial = torch.empty(max((2 if ctx.x0_amax is not None else 1) * sl * F * F0, 2 * sl * F * F if L > 1 else 0), **f32)
        pack = torch.empty(lib.sl_sage_stack_pack_bytes(n, L, arr), dtype=torch.uint8, device=dev)
        wbase, wstep = dW.data_ptr(), F * F * 4
        sbase, sstep, bbase = ds.data_ptr(), 2 * F * 4, db.data_ptr()
        for l in range(L):
            y = arr[l]
            if l == 0:
                y.dWs, y.dWn = dW0.data_ptr(), dW0.data_ptr() + F * F0 * 4
            else:
                y.dWs, y.dWn = wbase + 2 * (l - 1) * wstep, wbase + (2 * (l - 1) + 1) * wstep
            y.dscale, y.doffset = sbase + 2 * l * sstep, sbase + (2 * l + 1) * sstep
            y.dbias = (bbase + l * sstep) if (ctx.has[6 * l + 1] or ctx.has[6 * l + 3]) else None
        a = _adj_struct(adj, want_dx0 or L > 1)
        rows32 = rows.to(torch.int32) if rows is not None else None
        plan = ctx.plan
        top_g = None
        if (plan is not None and SPARSE_TOP_BWD and L >= 3 and rows is not None and F % 32 == 0 and float(ctx.meta_l[L - 1][1]) == 0.0
                and plan.matches(adj.csr, r) and plan.rowmap.numel() == n and lib.sl_gemm_act_norm_supported(F, F)
                and adj.csr.spmm_blocks[0] is not None and F >= BLOCKDIAG_MIN_F):
            # The read-out gradient lives on the roots: the top layer's backward pass on the rows R, the layer below it from the
            # compact rows T = R u N(R) (the passes of _SageDense._sparse_top_backward / _compact_dz_backward on this node's
            # buffers), then ONE C call for the dense chained passes of the layers below (sl_sage_stack_bwd_ready).
            top_g = _SageStack._sparse_top(ctx, lib, a, d, plan, params, X0, arr, ds, db, buf, am, chain_partial, tn_partial, pack, dX0)
        if top_g is None:
            check(lib.sl_sage_stack_bwd(C.byref(a), X0.data_ptr(), X0.stride(0), ctx.x0_amax.data_ptr() if ctx.x0_amax is not None else None, L,
                                        arr, d.data_ptr(), rows32.data_ptr() if rows32 is not None else None, r if rows32 is not None else 0,
                                        dX0.data_ptr() if dX0 is not None else None, buf.data_ptr(), am.data_ptr(), an_partial.data_ptr(),
                                        chain_partial.data_ptr() if chain_partial is not None else None, tn_partial.data_ptr(),
                                        pack.data_ptr(), _stream(X0)))
        _SageDense.chained_calls += L - 1
        ctx.keep = ctx.arr = None
        # (one unbind per buffer instead of an index op per gradient: ~40 views per step)
        Wg = dW0.unbind(0) + (dW.unbind(0) if L > 1 else ())
        sg, bg = ds.unbind(0), db.unbind(0)
        grads = []
        ng = ctx.needs_input_grad
        for l in range(L):
            hb_s, hb_n = ctx.has[6 * l + 1], ctx.has[6 * l + 3]
            k = 6 + 6 * l
            if top_g is not None and l in top_g:           # (the two row-sparse layers: their gradients are separate tensors)
                tWs, tWn, tbi, tsc, tof = top_g[l]
                grads += [tWs if ng[k] else None, tbi[0] if (hb_s and ng[k + 1] and tbi is not None) else None, tWn if ng[k + 2] else None,
                          tbi[1] if (hb_n and ng[k + 3] and tbi is not None) else None, tsc if ng[k + 4] else None, tof if ng[k + 5] else None]
                continue
            grads += [Wg[2 * l] if ng[k] else None, bg[2 * l] if (hb_s and ng[k + 1]) else None, Wg[2 * l + 1] if ng[k + 2] else None,
                      bg[2 * l + 1] if (hb_n and ng[k + 3]) else None, sg[2 * l] if ng[k + 4] else None, sg[2 * l + 1] if ng[k + 5] else None]
        return (dX0, None, None, None, None, None, *grads)

    @staticmethod
    def _sparse_top(ctx, lib, a, d, plan, params, X0, arr, ds, db, buf, am, chain_partial, tn_partial, pack, dX0):
        """Layers L - 1 (on the roots R) and L - 2 (from the compact rows T) of the stack, then sl_sage_stack_bwd_ready for the
        layers 0 .. L - 3.  Returns {layer: (dWs, dWn, dbias [2, F] or None, dscale [2, F], doffset [2, F])} for the two
        row-sparse layers; the others' gradients land in the node's flat buffers as in the dense pass."""
        L, adj = ctx.L, ctx.adj
        AX0, big, amax, stats = ctx.keep
        n, F = big.shape[1], big.shape[2]
        dev = big.device
        f32 = dict(dtype=torch.float32, device=dev)
        st = _stream(big[0])
        opt = lambda t_: t_.data_ptr() if t_ is not None else None
        top, mid, low = L - 1, L - 2, L - 3
        Zs = lambda l: big[3 * l]
        Zn = lambda l: big[3 * l + 1]
        out = lambda l: big[3 * l + 2]
        AX = lambda l: big[3 * L + l - 1]                    # (l >= 1)
        prm = lambda l: params[6 * l:6 * l + 6]              # Ws, bs, Wn, bn, scale, offset
        meta = ctx.meta_l
        # ---- top layer on the roots (ops._SageDense._sparse_top_backward)
        Ws, bs, Wn, bn, sc, of = prm(top)
        R = plan.rows64
        P = int(R.numel())
        onR = [torch.empty(P, F, **f32) for _ in range(4)]           # (the four saved tensors on the roots' rows: one launch)
        rows_multi([("gather", x, y) for x, y in zip((Zs(top), Zn(top), out(mid), AX(top)), onR)], R, P)
        (dZsR, dZnR), tsc, tof, tbi = _an_bwd(onR[:2], [bs, bn], (meta[top][0],) * 2, sc, of, F, 1.0,
                                              (d,), [True, True], bs is not None or bn is not None, (0.0, 0))
        g_top = (weight_grad(dZsR, onR[2], min_rows=ROOT_GEMM_MIN_ROWS), weight_grad(dZnR, onR[3], min_rows=ROOT_GEMM_MIN_ROWS), tbi, tsc, tof)
        GS = (mm_nt(dZnR, Wn.t(), min_rows=ROOT_GEMM_MIN_ROWS), mm_nt(dZsR, Ws.t(), min_rows=ROOT_GEMM_MIN_ROWS))
        dXT = torch.empty(plan.t, F, **f32)
        check(lib.sl_top_dx(GS[0].data_ptr(), GS[1].data_ptr(), F, plan.T32.data_ptr(), plan.slot.data_ptr(), plan.epos.data_ptr(),
                            plan.self_idx.data_ptr(), plan.targets32.data_ptr(), opt(adj.edge_w), opt(adj.row_scale), opt(adj.col_scale),
                            plan.t, F, dXT.data_ptr(), F, st))
        # ---- the layer below: act + norm backward on the rows T (through its output dropout mask), dZ compact
        Wsm, bsm, Wnm, bnm, scm, ofm = prm(mid)
        t = plan.t
        dZsT = torch.empty(t, F, **f32)
        dZnT = torch.empty(t + 1, F, **f32)
        dZnT[t].zero_()
        _dz, msc, mof, mbi = _an_bwd([Zs(mid), Zn(mid)], [bsm, bnm], (meta[mid][0],) * 2, scm, ofm, F, 1.0, (dXT,), [True, True],
                                     bsm is not None or bnm is not None, (float(meta[mid][1]), int(meta[mid][2])),
                                     dz_out=[dZsT, dZnT[:t]], row_idx=plan.T32, dz_compact=True)
        Tl = plan.T32.long()
        Xm = out(low) if mid >= 1 else X0
        onT = [torch.empty(t, Xm.shape[1], **f32), torch.empty(t, F, **f32)]
        rows_multi([("gather", Xm, onT[0]), ("gather", AX(mid), onT[1])], Tl, t)
        g_mid = (weight_grad(dZsT, onT[0]), weight_grad(dZnT[:t], onT[1]), mbi, msc, mof)
        # ---- its input gradient = the output gradient of layer L - 3: (A^T dZn) Wn + scatter_T(dZs[T] Ws), with that layer's act +
        #      norm backward in the product's epilogue (ops._SageDense._compact_dz_backward)
        AtdZn, amx = _at_dzn_on_rows(adj, plan, dZnT, n, F)
        corr = mm_nt(dZsT, Wsm.t())
        wpack = torch.empty(lib.sl_gemm_act_norm_pack_bytes(F, F), dtype=torch.uint8, device=dev)
        check(lib.sl_gemm_act_norm_pack_b2(Wnm.data_ptr(), 1, Wnm.stride(0), F, Wnm.data_ptr(), 1, Wnm.stride(0), F, F, wpack.data_ptr(), st))
        _Wl, bsl, _Wnl, bnl, scl, ofl = prm(low)
        low_buf = rows_empty(0, n, 3 * F, dev)
        low_amax = torch.empty(n, **f32)
        partial = torch.empty(lib.sl_gemm_an_bwd_partial_floats(n, F, 2), **f32)
        ld2 = (C.c_int64 * 2)(F, F)
        ld3 = (C.c_int64 * 2)(3 * F, 3 * F)
        ac = (C.c_int * 2)(meta[low][0], meta[low][0])
        dZb = (C.c_void_p * 2)(low_buf.data_ptr(), low_buf.data_ptr() + 8 * F)
        y = arr[low]
        nbytes = 4 * n * (F + 4 * F) + 4 * t * F
        with _timed(f"gemm_an_bwd_corr_nb2_N{F}", nbytes, dev, flops=2 * n * F * F):
            check(lib.sl_gemm_an_bwd_corr(AtdZn.data_ptr(), F, amx.data_ptr(), wpack.data_ptr(), n, F, F, 2, _ptr_array([Zs(low), Zn(low)]), ld2,
                                          _ptr_array([bsl, bnl]), ac, scl.data_ptr(), ofl.data_ptr(), 1.0, dZb, ld3,
                                          y.dscale, y.doffset, y.dbias, partial.data_ptr(), float(meta[low][1]), int(meta[low][2]),
                                          low_amax.data_ptr(), stats[low].data_ptr() if (stats is not None and y.row_stats) else None,
                                          corr.data_ptr(), corr.stride(0), plan.rowmap.data_ptr(), t, st))
        # ---- the layers 0 .. L - 3: dense chained passes, one C call
        check(lib.sl_sage_stack_bwd_ready(C.byref(a), X0.data_ptr(), X0.stride(0), ctx.x0_amax.data_ptr() if ctx.x0_amax is not None else None,
                                          low + 1, arr, low_buf.data_ptr(), low_amax.data_ptr(), dX0.data_ptr() if dX0 is not None else None,
                                          buf.data_ptr(), am.data_ptr(), opt(chain_partial), tn_partial.data_ptr(), pack.data_ptr(), st))
        _SageDense.sparse_top_calls += 1
        _SageDense.compact_dz_calls += 1
        _SageStack.sparse_top_calls += 1
        return {top: g_top, mid: g_mid}



def sage_stack_usable(mods) -> bool:
    """The static part of _SageStack's preconditions for a list of layers.GraphSAGE modules (everything that does not change
    from step to step; DeepGNN caches it)."""
    if not (SAGE_STACK and FUSED_LAYER_CALLS and GEMM_SPLIT and CHAIN_SAGE_BWD and ROOTS_SPARSE_GRAD and mods):
        return False
    F = mods[0].f_lin_self.weight.shape[0]
    if not (F % 32 == 0 and 32 <= F <= 256):
        return False
    for l, md in enumerate(mods):
        ws, wn = md.f_lin_self.weight, md.f_lin_neigh.weight
        Fi = ws.shape[1]
        if not (getattr(md, "norm", None) == "norm_feat" and md.act is None and md.act_name in ACT_CODE and tuple(ws.shape) == tuple(wn.shape)
                and ws.shape[0] == F and (Fi == F if l else (Fi % 4 == 0 and Fi <= 256)) and ws.is_cuda and ws.dtype == torch.float32
                and ws.stride(1) == 1 and wn.stride(1) == 1 and tuple(md.scale.shape) == (2, F) and md.scale.is_contiguous()
                and md.offset.is_contiguous() and md.scale.dtype == torch.float32):
            return False
    return True


class StepPath(NamedTuple):
    """How one conv stack of a step runs (step_path): ``forward`` / ``backward`` name a row of the table below."""
    forward: str
    backward: str


def step_path(kind: str, n: int, F: int, layers: int, training: bool, readout: str, stackable: bool = True, blockdiag: bool = True,
              heads: int = 1, residue: str = "none") -> StepPath:
    """THE table of the ways through a conv stack: (layer kind, batch rows n, hidden width F, layer count, training, read-out) -> path.
    models.DeepGNN._run_stack dispatches on it; tests/test_layers_gpu.py::test_step_path_table walks its cells (the entry counters
    must show the path the table names, the results must equal the kernel-by-kernel path's).

    ``stackable``: the stack's static preconditions hold (sage_stack_usable / gcn_stack_usable, residue 'none', node task, inner
    input dropouts fused into the producing layer); ``readout``: the pooling -- 'center' = one row per subgraph, 'mean' / 'max' / 'sum'
    = every row; ``residue``: 'none' = the read-out reads the LAST layer only, anything else = every layer's plain output (the lower
    layers are then dual-output: one tensor for the read-out, the dropped one for the next layer); ``blockdiag``: the batch carries
    its subgraph offsets (the block-diagonal aggregate).

      kind  rows n                       forward            backward (training)
      ----  ---------------------------  -----------------  ---------------------------------------------------------------
      any   n < GEMM_SPLIT_MIN_ROWS      kernels            kernels              torch.mm / SpMM / act_norm kernel by kernel
      sage  >= 1 024, stackable, center  stack              stack                sl_sage_stack_fwd / sl_sage_stack_bwd: one C call each
      sage  ... n >= SPARSE_TOP_BWD_MIN  stack              stack+sparse-top     two top layers on R / T = R u N(R), the rest
            _ROWS, >= 3 layers, F >= 96                                          sl_sage_stack_bwd_ready (the headline step)
      sage  ... fewer layers / F < 96    layer-calls        chained+sparse-top   _SageDense nodes: sl_sage_fwd per layer, ChainLink
      sage  >= 1 024, not stackable,     layer-calls        chained              sl_sage_fwd / sl_sage_bwd_chain per layer
            center
      sage  >= 1 024, residue none,      layer-calls        chained              the lower layers have ONE reader (the next layer): chained as
            pooled read-out                                                      under centre pooling, dense gradient into the top layer
      sage  >= 1 024, residue != none,   layer-calls        chained              dual-output layers (round 6): the plain output's gradient from
            pooled, 128 < F <= 256                                               the pooling node is added in the chained epilogue
      sage  >= 1 024, residue != none,   layer-calls        layer-calls          dual-output layers: sl_sage_fwd / sl_sage_bwd_chain
            centre pooling / other F                                             without a link between the layers
      gcn   >= 1 024, stackable, center  stack              stack                sl_gcn_stack_fwd / sl_gcn_stack_bwd
      gcn   >= 1 024, not stackable      layer-calls        layer-calls          sl_gcn_fwd / sl_gcn_bwd per layer
      gat   >= 1 024, F == 256,          pair-tail          dense | rows         sl_gemm_nt2_gat_f32 + sl_gat_fwd_rows; backward
            head width 4 * 2^k <= 128                                            row-sparse ('rows') from SPARSE_TOP_BWD_MIN_ROWS, center
      gat   >= 1 024, other widths       node-pass          dense | rows         sl_gemm_nt2_f32 + sl_gat_fwd (per-node pass)
    Evaluation (``training`` False): the forward column, backward 'none'."""
    tall = bool(GEMM_SPLIT and FUSED_LAYER_CALLS and n >= max(1, GEMM_SPLIT_MIN_ROWS))
    none = "none"
    if not tall:
        return StepPath("kernels", "kernels" if training else none)
    center = readout == "center" and residue == "none"
    big = bool(training and SPARSE_TOP_BWD and center and n >= SPARSE_TOP_BWD_MIN_ROWS)
    if kind == "sage":
        if stackable and center and SAGE_STACK:
            if big and not (SPARSE_TOP_STACK and layers >= 3 and F % 32 == 0 and F >= BLOCKDIAG_MIN_F and blockdiag):
                return StepPath("layer-calls", "chained+sparse-top")
            return StepPath("stack", ("stack+sparse-top" if big else "stack") if training else none)
        # (a read-out that reads every row of every layer -- mean / max / sort pooling, residue concat / max -- puts the layers in
        #  dual-output mode: no ChainLink between them, every layer's own one-call backward)
        # (round 6: a lower layer is dual-output only when the read-out reads it -- residue != 'none' -- and such layers chain too
        #  under a pooled read-out at widths in (128, 256]: CHAIN_DUAL, the plain output's gradient added in the epilogue above)
        if residue == "none":
            can_chain = True
        else:
            can_chain = bool(CHAIN_DUAL and 128 < F <= 256 and readout in ("mean", "max", "sum"))
        chained = "chained" if (CHAIN_SAGE_BWD and F % 32 == 0 and can_chain) else "layer-calls"
        return StepPath("layer-calls", ((chained + "+sparse-top") if (big and chained == "chained") else chained) if training else none)
    if kind == "gcn":
        if stackable and center and SAGE_STACK:
            return StepPath("stack", "stack" if training else none)
        return StepPath("layer-calls", "layer-calls" if training else none)
    if kind == "gat":
        D = F // max(1, heads)
        fwd = "pair-tail" if (GAT_PAIR_TAIL and F == 256 and D * heads == F and D % 4 == 0 and D <= 128 and (D // 4) & (D // 4 - 1) == 0) else "node-pass"
        return StepPath(fwd, ("rows" if big else "dense") if training else none)
    raise ValueError(f"step_path: unknown layer kind {kind!r}")


def sparse_top_stack_usable(csr, mods) -> bool:
    """Can _SageStack run the row-sparse backward of its two top layers on this batch?  (Three layers or more of one width F with
    F % 32 == 0 and the block-diagonal aggregate: what the compact-dZ pass of the layer below the top needs.  Otherwise the
    layer-by-layer nodes take the row-sparse pass, with their dense fall-back for that layer.)"""
    F = mods[0].f_lin_self.weight.shape[0]
    blocks = getattr(csr, "spmm_blocks", None)
    return bool(SPARSE_TOP_STACK and len(mods) >= 3 and F % 32 == 0 and F >= BLOCKDIAG_MIN_F and blocks is not None and blocks[0] is not None
                and _lib.load().sl_gemm_act_norm_supported(F, F))


def sage_stack(X0: torch.Tensor, adj: "NormAdj", mods, rows: Optional[torch.Tensor], plan=None):
    """out_L[rows] of the GraphSAGE modules ``mods`` applied in turn to X0 (already through layer 0's input dropout) -- see
    _SageStack; every layer's fused output dropout is ``mods[l]._out_p()`` (drawn here, in layer order).  Returns None when the
    stack form does not apply to this call (the caller then runs the layers one by one)."""
    n = X0.shape[0]
    if n < max(1, GEMM_SPLIT_MIN_ROWS) or not X0.is_cuda:
        return None
    params = []
    for md in mods:
        params += [md.f_lin_self.weight, md.f_lin_self.bias, md.f_lin_neigh.weight, md.f_lin_neigh.bias, md.scale, md.offset]
    if torch.is_grad_enabled() and not all(p is None or p.requires_grad for p in params):
        return None                                          # (frozen parameters: the layer-by-layer nodes handle them)
    X0 = _f32c(X0)
    if not (X0.stride(1) == 1 and X0.stride(0) % 4 == 0 and X0.data_ptr() % 16 == 0):
        X0 = X0.contiguous()
    F = mods[0].f_lin_self.weight.shape[0]
    meta = []
    for md in mods:
        drop = _drop_arg(md._out_p(), F)
        meta.append((ACT_CODE[md.act_name], float(drop[0]), int(drop[1])))
    return _SageStack.apply(X0, adj, rows, tuple(meta), torch.is_grad_enabled(), plan, *params)


class _GcnStack(torch.autograd.Function):
    """The conv loop of DeepGNN.forward over L GCN layers (shaDow/layers.py:417-444) plus the read-out's row select as ONE
    autograd node (sl_gcn_stack_fwd / sl_gcn_stack_bwd: the one-call layer passes of _GcnDense issued from C, identical
    results).  Same preconditions as _SageStack: nothing but layer l + 1 reads layer l's output, nothing but the row select
    the last one; one hidden width."""
    calls = 0

    @staticmethod
    def forward(ctx, X0, adj, rows, meta, grad_on, *params):
        lib = _lib.load()
        L = len(meta)
        n, F0 = X0.shape
        F = params[0].shape[0]
        dev = X0.device
        f32 = dict(dtype=torch.float32, device=dev)
        pitch0 = X0.stride(0) if (X0.stride(0) != F0 and X0.stride(0) % 32 == 0) else F0
        AX0 = torch.empty(n, pitch0, **f32)
        keep_all = bool(grad_on) and any(ctx.needs_input_grad)   # (no backward pass will come: one Z / A X slot, two `out` slots in turn)
        # per layer Z, out; then A X of the layers 1 .. L - 1 (training: one allocation per saved tensor, see STACK_SEPARATE_SLOTS)
        if keep_all and STACK_SEPARATE_SLOTS:
            big = _SlotList([rows_empty(0, n, F, dev) for _ in range(3 * L - 1)])
        else:
            big = torch.empty(3 * L - 1, n, F, **f32) if keep_all else torch.empty(4 if L > 1 else 2, n, F, **f32)
        slot_z = (lambda l: 2 * l) if keep_all else (lambda l: 0)
        slot_out = (lambda l: 2 * l + 1) if keep_all else (lambda l: 1 + (l & 1) if L > 1 else 1)
        slot_ax = (lambda l: 2 * L + l - 1) if keep_all else (lambda l: 3)
        arr = (_lib.SlGcnStackLayer * L)()
        sp = (lambda k: big[k].data_ptr()) if isinstance(big, _SlotList) else (lambda k, base=big.data_ptr(), step=n * F * 4: base + k * step)
        for l in range(L):
            W, b, sc, of = params[4 * l:4 * l + 4]
            y = arr[l]
            y.W, y.scale, y.offset, y.ldw = W.data_ptr(), sc.data_ptr(), of.data_ptr(), W.stride(0)
            y.b = b.data_ptr() if b is not None else None
            y.Fin, y.Fout = (F0 if l == 0 else F), F
            y.act, y.drop_p, y.drop_seed = meta[l]
            if l == 0:
                y.AX, y.ldax = AX0.data_ptr(), AX0.stride(0)
            else:
                y.AX, y.ldax = sp(slot_ax(l)), F
            y.Z, y.out = sp(slot_z(l)), sp(slot_out(l))
        pack = torch.empty(lib.sl_gcn_stack_pack_bytes(n, L, arr), dtype=torch.uint8, device=dev)
        a = _adj_struct(adj, False)
        check(lib.sl_gcn_stack_fwd(C.byref(a), X0.data_ptr(), X0.stride(0), L, arr, pack.data_ptr(), _stream(X0)))
        if Z_TAP is not None and keep_all:
            for l in range(L):
                _tap([big[2 * l]], [params[4 * l + 1]])
        ctx.save_for_backward(X0, *[p for p in params if p is not None])
        ctx.has = [p is not None for p in params]
        ctx.rows, ctx.adj, ctx.arr, ctx.L = rows, adj, arr, L
        ctx.keep = (AX0, big)
        ctx.set_materialize_grads(False)
        _GcnStack.calls += 1
        _GcnStack.last_slots = int(big.shape[0])
        fire_deferred()
        out = big[slot_out(L - 1)]
        return out.index_select(0, rows) if rows is not None else out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        saved = ctx.saved_tensors
        X0 = saved[0]
        it = iter(saved[1:])
        params = [next(it) if h else None for h in ctx.has]
        L, arr, adj, rows = ctx.L, ctx.arr, ctx.adj, ctx.rows
        if arr is None:
            raise RuntimeError("the layer-stack node was already back-propagated through (its forward products are released after the "
                               "first backward pass; retain_graph is not supported on this path: set SHADOW_SAGE_STACK=0)")
        n, F0 = X0.shape
        F = params[0].shape[0]
        dev = X0.device
        f32 = dict(dtype=torch.float32, device=dev)
        r = int(rows.numel()) if rows is not None else n
        d = _f32c(dout).contiguous() if dout is not None else torch.zeros(r, F, **f32)
        want_dx0 = bool(ctx.needs_input_grad[0])
        Fm = max(F, F0)
        dX0 = torch.empty(n, F0, **f32) if want_dx0 else None
        dW = torch.empty(max(1, L - 1), F, F, **f32)
        dW0 = torch.empty(F, F0, **f32)
        ds = torch.empty(3 * L, F, **f32)                    # per layer dscale, doffset, dbias
        grad = torch.empty(2, n, Fm, **f32)
        buf = torch.empty(n * (F + Fm), **f32)
        an_partial = torch.empty(2048 * 3 * F, **f32)
        tn_partial = torch.empty(lib.sl_gemm_tn_slices(n) * F * Fm, **f32)
        pack = torch.empty(lib.sl_gcn_stack_pack_bytes(n, L, arr), dtype=torch.uint8, device=dev)
        sbase, sstep = ds.data_ptr(), F * 4
        for l in range(L):
            y = arr[l]
            y.dW = dW0.data_ptr() if l == 0 else dW.data_ptr() + (l - 1) * F * F * 4
            y.dscale, y.doffset = sbase + 3 * l * sstep, sbase + (3 * l + 1) * sstep
            y.dbias = (sbase + (3 * l + 2) * sstep) if ctx.has[4 * l + 1] else None
        a = _adj_struct(adj, want_dx0 or L > 1)
        rows32 = rows.to(torch.int32) if rows is not None else None
        check(lib.sl_gcn_stack_bwd(C.byref(a), L, arr, d.data_ptr(), rows32.data_ptr() if rows32 is not None else None,
                                   r if rows32 is not None else 0, dX0.data_ptr() if dX0 is not None else None, grad.data_ptr(),
                                   buf.data_ptr(), an_partial.data_ptr(), tn_partial.data_ptr(), pack.data_ptr(), _stream(X0)))
        ctx.keep = ctx.arr = None
        Wg = (dW0,) + (dW.unbind(0) if L > 1 else ())
        sg = ds.unbind(0)
        ng = ctx.needs_input_grad
        grads = []
        for l in range(L):
            k = 5 + 4 * l
            grads += [Wg[l] if ng[k] else None, sg[3 * l + 2] if (ctx.has[4 * l + 1] and ng[k + 1]) else None,
                      sg[3 * l].view(params[4 * l + 2].shape) if ng[k + 2] else None,
                      sg[3 * l + 1].view(params[4 * l + 3].shape) if ng[k + 3] else None]
        return (dX0, None, None, None, None, *grads)


def gcn_stack_usable(mods) -> bool:
    """The static part of _GcnStack's preconditions for a list of layers.GCN modules."""
    if not (SAGE_STACK and FUSED_LAYER_CALLS and GEMM_SPLIT and mods):
        return False
    F = mods[0].f_lin.weight.shape[0]
    if not (F % 4 == 0 and 16 <= F <= 256):
        return False
    for l, md in enumerate(mods):
        w = md.f_lin.weight
        Fi = w.shape[1]
        if not (getattr(md, "norm", None) == "norm_feat" and md.act is None and md.act_name in ACT_CODE and w.shape[0] == F
                and (Fi == F if l else (Fi % 4 == 0 and Fi <= 256)) and w.is_cuda and w.dtype == torch.float32 and w.stride(1) == 1
                and md.scale.numel() == F and md.scale.is_contiguous() and md.offset.is_contiguous() and md.scale.dtype == torch.float32):
            return False
    return True


def gcn_stack(X0: torch.Tensor, adj: "NormAdj", mods, rows: Optional[torch.Tensor]):
    """out_L[rows] of the GCN modules ``mods`` applied in turn to X0 (already through layer 0's input dropout) -- see _GcnStack;
    None when the stack form does not apply to this call."""
    n = X0.shape[0]
    if n < max(1, GEMM_SPLIT_MIN_ROWS) or not X0.is_cuda or not isinstance(adj, NormAdj):
        return None
    params = []
    for md in mods:
        params += [md.f_lin.weight, md.f_lin.bias, md.scale, md.offset]
    if torch.is_grad_enabled() and not all(p is None or p.requires_grad for p in params):
        return None
    X0 = _f32c(X0)
    if not (X0.stride(1) == 1 and X0.stride(0) % 4 == 0 and X0.data_ptr() % 16 == 0):
        X0 = X0.contiguous()
    F = mods[0].f_lin.weight.shape[0]
    meta = []
    for md in mods:
        drop = _drop_arg(md._out_p(), F)
        meta.append((ACT_CODE[md.act_name], float(drop[0]), int(drop[1])))
    return _GcnStack.apply(X0, adj, rows, tuple(meta), torch.is_grad_enabled(), *params)


# The head of a node-classification step (L2 normalisation of the root embeddings, the one-layer classifier, softmax cross
# entropy) as one kernel forward and two backward (csrc/head.hip) instead of ~35 small torch / HIP kernels and their launches.
# SHADOW_FUSED_HEAD=0: the separate nodes.
FUSED_HEAD = os.environ.get("SHADOW_FUSED_HEAD", "1") != "0"

class _NodeHead(torch.autograd.Function):
    """(loss, preds, softmax(preds), normalised embeddings) of shaDow/models.py:200-203 + :163-166 for a one-layer classifier
    with feature normalisation: sl_head_fwd / sl_head_bwd.  Only ``loss`` carries a gradient."""
    calls = 0

    @staticmethod
    def forward(ctx, emb, W, b, scale, offset, label):
        lib = _lib.load()
        r, F = emb.shape
        Cn = W.shape[0]
        dev = emb.device
        f32 = dict(dtype=torch.float32, device=dev)
        xn = torch.empty(r, F, **f32)
        zpp = torch.empty(3, r, Cn, **f32)                   # z, preds, softmax(preds)
        small = torch.empty(2 * r + 1, **f32)                # |emb_i|, the roots' losses, the mean loss
        step = r * Cn * 4
        check(lib.sl_head_fwd(emb.data_ptr(), emb.stride(0), W.data_ptr(), W.stride(0), b.data_ptr() if b is not None else None,
                              scale.data_ptr(), offset.data_ptr(), label.data_ptr(), r, F, Cn, xn.data_ptr(), zpp.data_ptr(),
                              zpp.data_ptr() + step, zpp.data_ptr() + 2 * step, small.data_ptr(), small.data_ptr() + 4 * r,
                              small.data_ptr() + 8 * r, _stream(emb)))
        ctx.save_for_backward(W, scale, label)
        ctx.keep = (xn, zpp, small)
        ctx.shapes = (tuple(scale.shape), tuple(offset.shape), b is not None)
        ctx.set_materialize_grads(False)
        _NodeHead.calls += 1
        _z, preds, prob = zpp.unbind(0)
        loss = small[2 * r:].view(())
        ctx.mark_non_differentiable(preds, prob, xn)
        return loss, preds, prob, xn

    @staticmethod
    def backward(ctx, dloss, *_unused):
        lib = _lib.load()
        W, scale, label = ctx.saved_tensors
        if ctx.keep is None:
            raise RuntimeError("the fused head was already back-propagated through (retain_graph is not supported on this path: set "
                               "SHADOW_FUSED_HEAD=0)")
        xn, zpp, small = ctx.keep
        r, F = xn.shape
        Cn = W.shape[0]
        dev = xn.device
        f32 = dict(dtype=torch.float32, device=dev)
        if dloss is None:
            dloss = torch.zeros((), **f32)
        g = dloss.to(torch.float32).contiguous()
        demb = torch.empty(r, F, **f32)
        dW = torch.empty(Cn, F, **f32)
        dsm = torch.empty(3, Cn, **f32)                      # dbias, dscale, doffset
        work = torch.empty(3 * r * Cn, **f32)
        partial = torch.empty(max(1, int(lib.sl_head_partial_floats(r, F, Cn))), **f32)
        step = r * Cn * 4
        check(lib.sl_head_bwd(g.data_ptr(), xn.data_ptr(), zpp.data_ptr(), zpp.data_ptr() + 2 * step, small.data_ptr(), label.data_ptr(),
                              W.data_ptr(), W.stride(0), scale.data_ptr(), r, F, Cn, demb.data_ptr(), dW.data_ptr(), dsm.data_ptr(),
                              dsm.data_ptr() + 4 * Cn, dsm.data_ptr() + 8 * Cn, work.data_ptr(), partial.data_ptr(), _stream(xn)))
        ctx.keep = None
        sshape, oshape, has_b = ctx.shapes
        db, dsc, dof = dsm.unbind(0)
        ng = ctx.needs_input_grad
        return (demb if ng[0] else None, dW if ng[1] else None, db if (has_b and ng[2]) else None,
                dsc.view(sshape) if ng[3] else None, dof.view(oshape) if ng[4] else None, None)


def node_head_usable(emb, lin, scale, offset, label) -> bool:
    W = lin.weight
    return bool(FUSED_HEAD and torch.is_tensor(emb) and emb.is_cuda and emb.dim() == 2 and emb.dtype == torch.float32
                and emb.shape[0] > 0 and emb.shape[1] % 4 == 0 and emb.shape[1] <= 256 and W.shape[0] <= 256 and W.shape[1] == emb.shape[1]
                and W.dtype == torch.float32 and W.stride(1) == 1 and W.stride(0) % 4 == 0 and W.data_ptr() % 16 == 0
                and scale.numel() == W.shape[0] and scale.is_contiguous() and offset.is_contiguous() and scale.dtype == torch.float32
                and label is not None and label.dim() == 1 and label.dtype == torch.int64 and label.shape[0] == emb.shape[0]
                and label.is_cuda)


def node_head(emb, lin, scale, offset, label):
    """(mean CE loss, preds, softmax(preds), F.normalize(emb)) -- see _NodeHead."""
    emb = _f32c(emb)
    if not (emb.stride(1) == 1 and emb.stride(0) % 4 == 0 and emb.data_ptr() % 16 == 0):
        emb = emb.contiguous()
    return _NodeHead.apply(emb, lin.weight, lin.bias, scale, offset, label.contiguous())


# (mean, 1 / std) per row and branch handed from the forward GEMM epilogue to the chained backward epilogue (False: recomputed)
ROW_STATS_HANDOVER = True
# Chained GraphSAGE backward (sl_sage_bwd_chain).  (A module attribute, like the path switches below without an environment
# variable: tests and A/B scripts set them in the process -- round 5 cut the environment switchboard to the ones a user needs.)
CHAIN_SAGE_BWD = True


# Work the minibatch extractor wants issued right AFTER the first aggregation of a step has been enqueued (the prefetch of
# the next batch's sampler call: launched at once it shares the chip with the step's HBM-bound head -- feature gather and
# the layer-0 aggregation -- and all three slow down; launched here it runs beside the matrix-core-bound GEMMs).
_DEFERRED: dict = {}


def defer(key, fn):
    """One pending callable per ``key`` (a newer one replaces an unfired older one)."""
    _DEFERRED[key] = fn


# Where the deferred call is issued.  "body" (default): when the last GNN layer of the forward pass is enqueued -- the
# call then runs beside the read-out / classifier / loss kernels and the first small backward kernels, a stretch of ~15
# tiny kernels that leaves the chip nearly idle for about as long as the sampler pipeline takes; "fwd": at the end of
# DeepGNN.forward; "agg": after the step's first aggregation (beside the forward GEMMs).  Same box, products benchmark
# (round-3 A/B), step ms / sampler pipeline ms in the step / north-star HBM fraction:
#   immediate 10.27 / 0.25 / 0.41    agg 10.20 / 0.31-0.40 / 0.40-0.43    fwd 10.21 / 0.28 / 0.44    body 10.21 / 0.21 / 0.46
DEFER_POINT = "body"


def fire_deferred(point: str = "agg"):
    """Called by the aggregation nodes ("agg") and by DeepGNN.step after its forward pass ("fwd", which also fires whatever
    an "agg" point never reached, e.g. an MLP stack)."""
    if point != DEFER_POINT and point != "fwd":         # ("fwd" is the last point of a step: it fires whatever is left)
        return
    while _DEFERRED:
        _k, fn = _DEFERRED.popitem()
        fn()


class _GcnDense(torch.autograd.Function):
    """A whole GCN layer (shaDow/layers.py:417-444), out = norm(act((A X) W^T + b)), as ONE autograd node with ONE C call
    per direction (sl_gcn_fwd / sl_gcn_bwd): at the reference's own batch sizes the separate SpMM and Linear + act + norm
    nodes cost more host time than GPU time.  Only built when ``fusable`` holds; GCN.forward keeps the two-node path
    otherwise (same kernels, same order: identical results)."""
    calls = 0                # forward passes through the one-call entry (tests assert on it)

    @staticmethod
    def fusable(X, adj, W):
        Fo, Fi = W.shape
        return (FUSED_LAYER_CALLS and GEMM_SPLIT and torch.is_tensor(X) and X.is_cuda and X.shape[0] > 0
                and X.shape[0] >= GEMM_SPLIT_MIN_ROWS and Fo % 4 == 0 and Fo <= 256 and Fi % 4 == 0 and Fi <= 256
                and X.dtype == torch.float32 and X.stride(1) == 1 and X.stride(0) % 4 == 0 and X.data_ptr() % 16 == 0
                and W.stride(1) == 1 and W.dtype == torch.float32 and isinstance(adj, NormAdj))

    @staticmethod
    def forward(ctx, X, adj, W, b, scale, offset, act, drop):
        lib = _lib.load()
        n, Fi = X.shape
        Fo = W.shape[0]
        dev = X.device
        f32 = dict(dtype=torch.float32, device=dev)
        sc = scale.reshape(1, Fo).contiguous().float()
        of = offset.reshape(1, Fo).contiguous().float()
        bc = b.detach().contiguous() if b is not None else None
        pitch = X.stride(0) if (X.stride(0) != Fi and X.stride(0) % 32 == 0) else Fi
        AX = torch.empty(n, pitch, **f32)[:, :Fi]
        Z, out = torch.empty(n, Fo, **f32), torch.empty(n, Fo, **f32)
        out2 = torch.empty(n, Fo, **f32) if _is_dual(drop) else None
        pack = torch.empty(lib.sl_gcn_pack_bytes(n, Fi, Fo), dtype=torch.uint8, device=dev)
        a = _adj_struct(adj, False)
        opt = lambda t: t.data_ptr() if t is not None else None
        check(lib.sl_gcn_fwd(C.byref(a), X.data_ptr(), X.stride(0), Fi, Fo, W.data_ptr(), W.stride(0), opt(bc), sc.data_ptr(),
                             of.data_ptr(), int(act), float(drop[0]), int(drop[1]), AX.data_ptr(), AX.stride(0), Z.data_ptr(),
                             out.data_ptr(), opt(out2), pack.data_ptr(), _stream(X)))
        _tap([Z], [bc])
        _GcnDense.calls += 1
        ctx.save_for_backward(AX, W, Z, sc, of, bc if bc is not None else sc.new_empty(0))
        ctx.adj = adj
        ctx.meta = (act, drop, scale.shape, offset.shape, b is not None, Fi)
        ctx.set_materialize_grads(False)
        fire_deferred()
        return out if out2 is None else (out, out2)

    @staticmethod
    def backward(ctx, *dout):
        lib = _lib.load()
        AX, W, Z, sc, of, b = ctx.saved_tensors
        act, drop, sshape, oshape, has_b, Fi = ctx.meta
        n, Fo = Z.shape
        dev = Z.device
        ng = ctx.needs_input_grad
        d0 = _f32c(dout[0]).contiguous() if dout[0] is not None else None
        d1 = None
        if _is_dual(drop):
            d1 = _f32c(dout[1]).contiguous() if dout[1] is not None else None
            if d1 is None:
                drop = (0.0, 0)
        if d0 is None and d1 is None:
            d0 = torch.zeros(n, Fo, dtype=torch.float32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        dX = torch.empty(n, Fi, **f32) if ng[0] else None
        dW = torch.empty(Fo, Fi, **f32)
        dbi = torch.empty(1, Fo, **f32) if has_b else None
        dsc, dof = torch.empty(1, Fo, **f32), torch.empty(1, Fo, **f32)
        buf = torch.empty(n * (Fo + Fi), **f32)
        an_partial = torch.empty(2048 * 1 * 3 * Fo, **f32)
        tn_partial = torch.empty(lib.sl_gemm_tn_slices(n) * Fo * Fi, **f32)
        pack = torch.empty(lib.sl_gcn_pack_bytes(n, Fi, Fo), dtype=torch.uint8, device=dev)
        a = _adj_struct(ctx.adj, bool(ng[0]))
        opt = lambda t: t.data_ptr() if t is not None else None
        check(lib.sl_gcn_bwd(C.byref(a), AX.data_ptr(), AX.stride(0), Z.data_ptr(), Fi, Fo, W.data_ptr(), W.stride(0),
                             opt(b if has_b else None), sc.data_ptr(), of.data_ptr(), int(act), float(drop[0]), int(drop[1]), opt(d0),
                             opt(d1), opt(dX), Fi, dW.data_ptr(), opt(dbi), dsc.data_ptr(), dof.data_ptr(), buf.data_ptr(),
                             an_partial.data_ptr(), tn_partial.data_ptr(), pack.data_ptr(), _stream(Z)))
        return (dX, None, dW if ng[2] else None, dbi[0] if (has_b and ng[3]) else None, dsc.reshape(sshape), dof.reshape(oshape),
                None, None)


def gcn_dense(X: torch.Tensor, adj: "NormAdj", lin, act: str, scale: torch.Tensor, offset: torch.Tensor,
              out_dropout: float = 0.0, dual: bool = False):
    """norm(act(lin(adj @ X))) -- GCN.forward (shaDow/layers.py:431-435) through the one-call entries.  The caller checks
    ``_GcnDense.fusable`` first."""
    if act not in ACT_CODE:
        raise NotImplementedError(f"activation {act!r} is not available in the fused HIP kernel")
    F = lin.weight.shape[0]
    return _GcnDense.apply(X, adj, lin.weight, lin.bias, scale, offset, ACT_CODE[act], _drop_arg(out_dropout, F, None, dual))


def sage_dense(X, adj: "NormAdj", lin_self, lin_neigh, act: str, scale: torch.Tensor,
               offset: torch.Tensor, out_dropout: float = 0.0, dual: bool = False, in_dropout: float = 0.0,
               chain_next: bool = False, roots_only: bool = False, pool_only: bool = False):
    """norm(act(lin_self(X))) + norm(act(lin_neigh(adj @ X))) -- GraphSAGE.forward (shaDow/layers.py:471-483).
    ``dual`` (with out_dropout > 0): returns (out, dropout(out)) from one kernel pass.  ``X`` may be a LazyRows
    (layer 0 of the fast path): gather and input dropout ``in_dropout`` then happen inside the aggregation kernel.
    ``chain_next``: the caller guarantees that ONLY the next GraphSAGE layer reads the returned tensor (see ChainLink);
    ``roots_only``: ... that only a row-selecting read-out reads it (ops.select_roots, see RootsLink); ``pool_only``: ... that only a
    pooled read-out does (ops.pool_and_roots: its gradient may then arrive as a table, POOL_GRAD_TABLE)."""
    if act not in ACT_CODE:
        raise NotImplementedError(f"activation {act!r} is not available in the fused HIP kernel")
    F = lin_self.weight.shape[0]
    code = ACT_CODE[act]
    lazy = None
    if isinstance(X, LazyRows):
        if FUSE_GATHER_INTO_SPMM and can_fuse_gather(adj, X):
            lazy, X = X, X.table.new_empty(0)
        else:
            X, _seed = X.gather_dropped(in_dropout)
    # chaining: a producer node left its ChainLink on the tensor it returned; ``chain_next`` asks this node to do the same
    link_down = getattr(X, "_shadow_chain", None) if torch.is_tensor(X) else None
    link_up = ChainLink() if (chain_next and (not dual or CHAIN_DUAL) and CHAIN_SAGE_BWD) else None
    if link_up is None and pool_only and not dual and not chain_next and POOL_GRAD_TABLE and CHAIN_DUAL and CHAIN_SAGE_BWD and 128 < F <= 256:
        link_up = ChainLink()
        link_up.plain_only = True
    link_roots = RootsLink() if (roots_only and not dual and not chain_next and ROOTS_SPARSE_GRAD) else None
    res = _SageDense.apply(X, adj, lin_self.weight, lin_self.bias, lin_neigh.weight, lin_neigh.bias, scale, offset,
                           (code, code), _drop_arg(out_dropout, F, None, dual), lazy, float(in_dropout), link_down, link_up, link_roots)
    if link_up is not None and link_up.published and link_up.plain_only and torch.is_tensor(res):
        res._shadow_plain = link_up
    elif link_up is not None and link_up.published and torch.is_tensor(res):
        res._shadow_chain = link_up
    elif link_up is not None and link_up.published and isinstance(res, tuple):
        # dual mode: the dropped output goes to the layer above (which finds the link on it), the plain one to the read-out
        # (ops.pool_and_roots leaves its gradient on the link)
        res[1]._shadow_chain = link_up
        res[0]._shadow_plain = link_up
    if link_roots is not None and link_roots.published and torch.is_tensor(res):
        res._shadow_roots = link_roots
    return res


def _drop_arg(out_dropout: float, F: int, seg: Optional[int] = None, dual: bool = False):
    """(p, seed[, True]) of the fused output dropout, or (0, 0)."""
    if out_dropout and out_dropout > 0.0:
        if not can_fuse_out_dropout(F, seg):
            raise ValueError(f"fused output dropout is not available for width {F}")
        return (float(out_dropout), new_dropout_seed(), True) if dual else (float(out_dropout), new_dropout_seed())
    if dual:
        raise ValueError("dual output needs out_dropout > 0")
    return (0.0, 0)


def linear_act_norm(Xs: List[torch.Tensor], lins: Sequence["torch.nn.Linear"], acts: Sequence[str],
                    scale: torch.Tensor, offset: torch.Tensor, seg: Optional[int] = None,
                    out_scale: float = 1.0, out_dropout: float = 0.0, dual: bool = False):
    """Fused dense tail of a layer: sum_b norm_b(act_b(lin_b(X_b))) * out_scale
    (nn.Linear + act + _f_norm_feat + add; shaDow/layers.py:434-435, :476-483, :393-394)."""
    codes = []
    for a in acts:
        if a not in ACT_CODE:
            raise NotImplementedError(f"activation {a!r} is not available in the fused HIP kernel")
        codes.append(ACT_CODE[a])
    F = lins[0].weight.shape[0]
    nb = len(Xs)
    return _LinearActNorm.apply(scale, offset, tuple(codes), int(seg or F), float(out_scale), nb, _drop_arg(out_dropout, F, seg, dual),
                                *Xs, *[l.weight for l in lins], *[l.bias for l in lins])


def act_norm(Zs: List[torch.Tensor], acts: Sequence[str], scale: torch.Tensor, offset: torch.Tensor,
             seg: Optional[int] = None, out_scale: float = 1.0, out_dropout: float = 0.0, dual: bool = False):
    """out_scale * sum_b norm_b(act_b(Z_b)) with the reference's 'norm_feat'
    (shaDowLayer._f_norm_feat, shaDow/layers.py:329-338).  scale/offset hold one
    row of F features per branch (any shape with nb*F elements)."""
    F = Zs[0].shape[1]
    codes = []
    for a in acts:
        if a not in ACT_CODE:
            raise NotImplementedError(f"activation {a!r} is not available in the fused HIP kernel "
                                      f"(supported: {sorted(ACT_CODE)})")
        codes.append(ACT_CODE[a])
    return _ActNorm.apply(scale, offset, tuple(codes), int(seg or F), float(out_scale), _drop_arg(out_dropout, F, seg, dual), *Zs)


# ----------------------------------------------------------------------------- readout / encodings
POOL_MODE = {"mean": 0, "max": 1, "sum": 2}


class _SegmentPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, node_off, mode):
        X = _f32c(X)
        _need_cuda(X, node_off)
        P, F = int(node_off.numel()) - 1, int(X.shape[1])
        out = torch.empty(P, F, dtype=torch.float32, device=X.device)
        am = torch.empty(P, F, dtype=torch.int32, device=X.device) if mode == 1 else None
        with _timed(f"segment_pool_F{F}", 4 * X.shape[0] * F + 4 * P * F, X.device):
            check(_lib.load().sl_segment_pool_fwd(X.data_ptr(), X.stride(0), node_off.data_ptr(), P, F, mode,
                                                  out.data_ptr(), out.stride(0),
                                                  am.data_ptr() if am is not None else None, _stream(X)))
        ctx.mode, ctx.n = mode, int(X.shape[0])
        ctx.save_for_backward(node_off, am if am is not None else node_off)
        return out

    @staticmethod
    def backward(ctx, dout):
        node_off, am = ctx.saved_tensors
        dout = _f32c(dout)
        P, F = dout.shape
        dX = torch.empty(ctx.n, F, dtype=torch.float32, device=dout.device)
        check(_lib.load().sl_segment_pool_bwd(dout.data_ptr(), dout.stride(0), node_off.data_ptr(), P, F, ctx.mode,
                                              am.data_ptr() if ctx.mode == 1 else None, dX.data_ptr(), dX.stride(0),
                                              _stream(dout)))
        return dX, None, None


class _PoolAndRoots(torch.autograd.Function):
    """(segment_pool(X), X[rows]) as ONE node: a read-out that pools every layer output AND reads its root rows
    (shaDow/layers.py:154-199 with mean / max / sum pooling) would otherwise hand autograd two dense [n, F] gradients per
    layer -- a zero-filled index_put and the pooling backward -- plus the pass that adds them.  Here the pooling backward
    writes the dense gradient once and the few root rows are added in place."""
    calls = 0
    table_calls = 0          # backward passes that handed a gradient TABLE to the chain (POOL_GRAD_TABLE)

    @staticmethod
    def forward(ctx, X, node_off, rows, mode, link=None):
        X = _f32c(X)
        _need_cuda(X, node_off, rows)
        ctx.link = link
        P, F = int(node_off.numel()) - 1, int(X.shape[1])
        out = torch.empty(P, F, dtype=torch.float32, device=X.device)
        am = torch.empty(P, F, dtype=torch.int32, device=X.device) if mode == 1 else None
        with _timed(f"segment_pool_F{F}", 4 * X.shape[0] * F + 4 * P * F, X.device):
            check(_lib.load().sl_segment_pool_fwd(X.data_ptr(), X.stride(0), node_off.data_ptr(), P, F, mode,
                                                  out.data_ptr(), out.stride(0),
                                                  am.data_ptr() if am is not None else None, _stream(X)))
        ctx.mode, ctx.n = mode, int(X.shape[0])
        ctx.save_for_backward(node_off, am if am is not None else node_off, rows)
        ctx.set_materialize_grads(False)
        # (what the nodes of one read-out share in their backward passes: the row -> table-row map of POOL_GRAD_TABLE, built once)
        ctx.shared = getattr(node_off, "_shd_pool_shared", None)
        if ctx.shared is None:
            ctx.shared = {}
            try:
                node_off._shd_pool_shared = ctx.shared
            except Exception:
                pass
        _PoolAndRoots.calls += 1
        return out, X.index_select(0, rows)

    @staticmethod
    def backward(ctx, dout, droots):
        node_off, am, rows = ctx.saved_tensors
        P = int(node_off.numel()) - 1
        F = int((dout if dout is not None else droots).shape[1])
        dev = node_off.device
        link = ctx.link
        hand_over = link is not None and link.published and link.dual and CHAIN_DUAL and link.plain_grad is None
        if (link is not None and link.published and link.plain_only and link.plain_grad is None
                and not (POOL_GRAD_TABLE and CHAIN_DUAL and ctx.mode != 1 and P > 0 and 128 < F <= 256 and F % 4 == 0)):
            link = None                              # (the last layer's link only ever carries a table)
        elif link is not None and link.plain_only:
            hand_over = link.published and link.plain_grad is None
        if hand_over and POOL_GRAD_TABLE and ctx.mode != 1 and P > 0 and 128 < F <= 256 and F % 4 == 0:
            # mean / sum pooling: one gradient row per subgraph (+ one per root) -- the chained layer above reads the table through a
            # row map instead of an [n, F] tensor this node would write and it would read (sl_pool_grad_table)
            lib = _lib.load()
            K = int(rows.numel())
            key = ("index", rows.data_ptr(), K, ctx.n)
            index = ctx.shared.get(key)
            if index is None:
                index = torch.empty(max(ctx.n, 1), dtype=torch.int32, device=dev)[:ctx.n]
                check(lib.sl_pool_grad_rows(node_off.data_ptr(), P, rows.data_ptr(), K, ctx.n, index.data_ptr(), _stream(node_off)))
                ctx.shared[key] = index
            d_o = _f32c(dout).contiguous() if dout is not None else None
            d_r = _f32c(droots).contiguous() if droots is not None else None
            table = torch.empty(P + K, F, dtype=torch.float32, device=dev)
            check(lib.sl_pool_grad_table(d_o.data_ptr() if d_o is not None else None, F, d_r.data_ptr() if d_r is not None else None, F,
                                         node_off.data_ptr(), P, rows.data_ptr(), K, ctx.n, F, ctx.mode, table.data_ptr(), F,
                                         _stream(node_off)))
            link.plain_grad, link.plain_index = table, index
            link.plain_dummy = placeholder(ctx.n, F, dev)
            _PoolAndRoots.table_calls += 1
            return link.plain_dummy, None, None, None, None
        if dout is not None:
            dout = _f32c(dout)
            dX = torch.empty(ctx.n, F, dtype=torch.float32, device=dev)
            check(_lib.load().sl_segment_pool_bwd(dout.data_ptr(), dout.stride(0), node_off.data_ptr(), P, F, ctx.mode,
                                                  am.data_ptr() if ctx.mode == 1 else None, dX.data_ptr(), dX.stride(0),
                                                  _stream(dout)))
        else:
            dX = torch.zeros(ctx.n, F, dtype=torch.float32, device=dev)
        if droots is not None:
            dX.index_add_(0, rows, droots.to(dX.dtype))       # (the roots of a batch are distinct rows: no two adds meet)
        if hand_over:
            # X is the plain output of a dual-output GraphSAGE layer that chains its backward pass: the gradient stays on the link
            # (the layer above adds it in its epilogue, or the layer itself picks it up), autograd carries a storage-less placeholder
            link.plain_grad = dX
            link.plain_dummy = placeholder(ctx.n, F, dev)
            return link.plain_dummy, None, None, None, None
        return dX, None, None, None, None


def pool_and_roots(X: torch.Tensor, node_off: torch.Tensor, rows: torch.Tensor, mode: str):
    """(segment_pool(X, node_off, mode), X[rows]) with one dense gradient (see _PoolAndRoots)."""
    assert node_off.dtype == torch.int32 and rows.dtype == torch.int64
    return _PoolAndRoots.apply(X, node_off, rows, POOL_MODE[mode], getattr(X, "_shadow_plain", None))


def segment_pool(X: torch.Tensor, node_off: torch.Tensor, mode: str) -> torch.Tensor:
    """mean / max / sum of X over the rows of each subgraph; node_off = [P+1] int32 row offsets
    (F.embedding_bag over the subgraph offsets, shaDow/layers.py:166-183).  Rows must tile
    [0, n): every row belongs to exactly one subgraph (true for a collated batch)."""
    assert node_off.dtype == torch.int32
    return _SegmentPool.apply(X, node_off, POOL_MODE[mode])


ENC_KIND = {"hops": 0, "pprs": 1, "drnls": 2}


def encode_codes(kind: str, src: torch.Tensor, dim: int) -> torch.Tensor:
    """Bit mask of the active one-hot columns per node (frontend/graph.py:134-172)."""
    _need_cuda(src)
    assert src.dtype in (torch.int32, torch.float32) and src.dim() == 1
    src = src.contiguous()
    codes = torch.empty(max(1, src.numel()), dtype=torch.int32, device=src.device)[:src.numel()]
    check(_lib.load().sl_encode_codes(ENC_KIND[kind], src.data_ptr(), src.numel(), dim, codes.data_ptr(), _stream(src)))
    return codes


class OneHotCodes:
    """A one-hot (multi-hot at ppr bin edges) encoding kept as one bit mask per node; what the fast
    minibatch path puts into ``feat_aug_ens`` instead of the dense [n, dim] matrix."""

    def __init__(self, codes: torch.Tensor, dim: int):
        self.codes, self.dim = codes, int(dim)

    @property
    def shape(self):
        return (int(self.codes.numel()), self.dim)

    def dense(self) -> torch.Tensor:
        return codes_to_dense(self.codes, self.dim)


def codes_to_dense(codes: torch.Tensor, dim: int) -> torch.Tensor:
    """The [n, dim] fp32 one-hot matrix the reference materialises (tests / generic consumers)."""
    bits = torch.arange(dim, device=codes.device, dtype=torch.int32)
    return ((codes.unsqueeze(1) >> bits) & 1).to(torch.float32)


class _OnehotLinear(torch.autograd.Function):
    PARTIAL_BLOCKS = 512

    @staticmethod
    def forward(ctx, X, codes, weight, bias):
        X = _f32c(X)
        _need_cuda(X, codes, weight)
        n, F = X.shape
        dim = int(weight.shape[1])
        Wt = weight.detach().t().contiguous()                       # [dim, F]
        out = torch.empty(n, F, dtype=torch.float32, device=X.device)
        check(_lib.load().sl_onehot_linear_fwd(X.data_ptr(), X.stride(0), codes.data_ptr(), Wt.data_ptr(),
                                               bias.data_ptr() if bias is not None else None, n, F, dim,
                                               out.data_ptr(), out.stride(0), _stream(X)))
        ctx.save_for_backward(codes)
        ctx.dim, ctx.has_bias = dim, bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        (codes,) = ctx.saved_tensors
        dout = _f32c(dout)
        n, F = dout.shape
        dim = ctx.dim
        dWt = torch.empty(dim, F, dtype=torch.float32, device=dout.device)
        db = torch.empty(F, dtype=torch.float32, device=dout.device) if ctx.has_bias else None
        nblk = _OnehotLinear.PARTIAL_BLOCKS
        partial = torch.empty(nblk * (dim + 1) * F, dtype=torch.float32, device=dout.device)
        check(_lib.load().sl_onehot_linear_bwd(dout.data_ptr(), dout.stride(0), codes.data_ptr(), n, F, dim,
                                               dWt.data_ptr(), db.data_ptr() if db is not None else None,
                                               partial.data_ptr(), nblk, _stream(dout)))
        return dout, None, dWt.t(), db


def onehot_linear_add(X: torch.Tensor, codes: torch.Tensor, lin: "torch.nn.Linear") -> torch.Tensor:
    """X + lin(onehot(codes))  (the 'sum' feature augmentation of DeepGNN.forward) in one pass,
    without the [n, dim] one-hot matrix; lin.weight is [F, dim]."""
    return _OnehotLinear.apply(X, codes, lin.weight, lin.bias)
