"""Host-side sampler interfaces over libshadow_hip.so.

Two surfaces:

* ``HipSampler`` / ``DeviceBatch`` -- the fast path: a sampler call returns ONE
  block-diagonal batch as torch tensors resident in HBM (no per-subgraph host
  objects).  This is what ``minibatch.py`` and ``bench.py`` use.

* ``ParallelSampler`` / ``SubgraphStructVec`` -- drop-in mirror of the
  reference's pybind11 module ``ParallelSampler``
  (para_graph_sampler/graph_engine/backend/ParallelSampler.cpp:707-746): same
  class names, constructor arguments, method names, config keys and getter
  names, so ``frontend/samplers_ensemble.py``-style callers work unchanged.

There is no CPU fallback; everything routes through the HIP library.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import SG_AUG, SG_METHOD, CapacityError, SgBatchCounts, SgBatchOut, SgConfig, check


def _parse_bool(cfg: Dict[str, str], key: str, default: bool) -> bool:
    # ParallelSampler::_extract_bool_config, ParallelSampler.cpp:483-496
    if key in cfg:
        return str(cfg[key]) in ("true", "True", "1")
    return default


@dataclass
class SamplerConfig:
    """Parsed form of the reference's string->string sampler config dict
    (frontend/samplers_cpp.py:40-45,73-80,124-131)."""
    method: str = "khop"
    num_roots: int = 1
    depth: int = 2
    budget: int = -1
    k: int = 0
    threshold: float = 0.0
    add_self_edge: bool = False
    include_target_conn: bool = False
    compat_overread: bool = False
    aug: Sequence[str] = ()
    return_target_only: bool = False

    @classmethod
    def from_cpp_dict(cls, cfg: Dict[str, str], aug: Iterable[str] = ()):
        if "method" not in cfg:
            # the reference prints this and calls exit(1) (.cpp:676-679)
            raise KeyError("[HIP parallel sampler]: need to have the 'method' key in the config")
        m = cfg["method"]
        if m == "ppr_st":
            raise NotImplementedError(
                "method 'ppr_st' (ParallelSampler::ppr_stochastic) is not provided by the HIP backend")
        if m not in SG_METHOD:
            raise KeyError(f"unknown sampler method {m!r}")
        c = cls(method=m, num_roots=int(cfg["num_roots"]), aug=tuple(sorted(aug)))
        if m == "khop":
            c.depth = int(cfg["depth"])        # std::stoi(config.at(...)), .cpp:515-516
            c.budget = int(cfg["budget"])
        if m == "ppr":
            c.k = int(cfg["k"])                # .cpp:570-571
            c.threshold = float(cfg["threshold"])
        c.add_self_edge = _parse_bool(cfg, "add_self_edge", False)
        c.include_target_conn = _parse_bool(cfg, "include_target_conn", False)
        c.return_target_only = _parse_bool(cfg, "return_target_only", False)
        c.compat_overread = _parse_bool(cfg, "compat_overread", False)
        return c

    def to_c(self) -> SgConfig:
        flags = 0
        for a in self.aug:
            flags |= SG_AUG[a]
        return SgConfig(SG_METHOD[self.method], self.num_roots, self.depth, self.budget, self.k,
                        self.threshold, int(self.add_self_edge), int(self.include_target_conn),
                        int(self.compat_overread), flags)


@dataclass
class DeviceBatch:
    """One sampler call: P subgraphs in block-diagonal form, resident in HBM.

    int32 tensors hold uint32 bit patterns (node ids < 2^31 always; edge ids of
    graphs with more than 2^31 edges must be reinterpreted as unsigned)."""
    node: torch.Tensor        # [n]   original node id
    indptr: torch.Tensor      # [n+1] batch CSR row pointers
    indices: torch.Tensor     # [e]   batch column ids
    edge_id: torch.Tensor     # [e]   full-graph edge index (-1 = inserted self edge)
    target: torch.Tensor      # [P*num_roots] batch ids of the roots
    subg_node_off: torch.Tensor  # [P+1]
    subg_edge_off: torch.Tensor  # [P+1]
    ppr: torch.Tensor         # [n] float32
    hop: Optional[torch.Tensor] = None
    drnl: Optional[torch.Tensor] = None
    num_subgraphs: int = 0
    num_roots: int = 1
    counts: dict = field(default_factory=dict)

    @property
    def num_nodes(self):
        return int(self.node.shape[0])

    @property
    def num_edges(self):
        return int(self.indices.shape[0])

    @property
    def size_subg(self):
        return self.subg_node_off[1:] - self.subg_node_off[:-1]

    def to_host(self):
        """numpy copies with the unsigned interpretation."""
        def u(t):
            return t.detach().cpu().numpy().view(np.uint32)
        d = dict(node=u(self.node), indptr=u(self.indptr), indices=u(self.indices),
                 edge_id=u(self.edge_id), target=u(self.target),
                 subg_node_off=u(self.subg_node_off), subg_edge_off=u(self.subg_edge_off),
                 ppr=self.ppr.detach().cpu().numpy())
        d["hop"] = u(self.hop) if self.hop is not None else None
        d["drnl"] = u(self.drnl) if self.drnl is not None else None
        return d

    def split_host(self):
        """Per-subgraph LOCAL arrays in the reference's getter convention."""
        h = self.to_host()
        no, eo = h["subg_node_off"].astype(np.int64), h["subg_edge_off"].astype(np.int64)
        R = self.num_roots
        out = []
        for p in range(self.num_subgraphs):
            n0, n1, e0, e1 = no[p], no[p + 1], eo[p], eo[p + 1]
            d = dict(indptr=h["indptr"][n0:n1 + 1].astype(np.int64) - e0,
                     indices=h["indices"][e0:e1].astype(np.int64) - n0,
                     node=h["node"][n0:n1].astype(np.int64),
                     edge_index=h["edge_id"][e0:e1].astype(np.int64),
                     target=h["target"][p * R:(p + 1) * R].astype(np.int64) - n0,
                     ppr=h["ppr"][n0:n1])
            if h["hop"] is not None:
                d["hop"] = h["hop"][n0:n1].astype(np.int64)
            if h["drnl"] is not None:
                d["drnl"] = h["drnl"][n0:n1].astype(np.int64)
            out.append(d)
        return out


class _Pending:
    __slots__ = ("cfg", "P", "root_start", "serial", "bufs", "out", "roots_dev", "sizes")


class HipSampler:
    """Device-resident sampler (fast path).  One instance owns one sg_sampler
    handle: the full-graph CSR in HBM, the epoch's root list, the PPR table,
    scratch memory and the RNG serial counter."""

    def __init__(self, indptr=None, indices=None, *, device: Optional[torch.device] = None,
                 seed: int = -1, path_indptr: str = "", path_indices: str = "", bin_dtype=("uint32", "uint32")):
        """``bin_dtype``: element types of the two .bin files (uint32 as the reference writes them; 'int64' /
        'uint64' files are narrowed on the way in, values >= 2^32 are refused)."""
        self._lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise ValueError("HipSampler needs a ROCm device (torch device type 'cuda')")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        h = C.c_void_p()
        self._keep = None
        n_ptr = 0 if indptr is None else len(indptr)
        if n_ptr == 0:
            # empty arrays + paths => read the raw uint32 .bin files (ParallelSampler.h:41-46)
            wp, wx = (np.dtype(d).itemsize for d in bin_dtype)
            check(self._lib.sg_create_from_bin_ex(path_indptr.encode(), path_indices.encode(), wp, wx,
                                                  self.device.index, seed, C.byref(h)))
        elif isinstance(indptr, torch.Tensor) and indptr.is_cuda:
            assert indptr.dtype == torch.int32 and indices.dtype == torch.int32
            self._keep = (indptr.contiguous(), indices.contiguous())
            check(self._lib.sg_create(self._keep[0].data_ptr(), self._keep[1].data_ptr(),
                                      self._keep[0].numel() - 1, self._keep[1].numel(), 1,
                                      self.device.index, seed, C.byref(h)))
        else:
            ip = np.ascontiguousarray(indptr, dtype=np.uint32)
            ix = np.ascontiguousarray(indices, dtype=np.uint32)
            check(self._lib.sg_create(ip.ctypes.data, ix.ctypes.data, ip.size - 1, ix.size, 0,
                                      self.device.index, seed, C.byref(h)))
        self._h = h
        self._pending: Optional[_Pending] = None
        self._hwm = {}       # per sampler config: high-water marks of (edges, nodes) per subgraph, for output sizing

    # ------------------------------------------------------------------ info
    def close(self):
        if getattr(self, "_h", None):
            self._lib.sg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_nodes(self):
        return int(self._lib.sg_num_nodes(self._h))

    def num_edges(self):
        return int(self._lib.sg_num_edges(self._h))

    def num_nodes_target(self):
        return int(self._lib.sg_num_nodes_target(self._h))

    def get_idx_root(self):
        return int(self._lib.sg_get_idx_root(self._h))

    # --------------------------------------------------------------- targets
    def shuffle_targets(self, targets):
        t = np.ascontiguousarray(np.asarray(targets).reshape(-1), dtype=np.uint32)
        check(self._lib.sg_shuffle_targets(self._h, t.ctypes.data, t.size))

    def next_roots(self, num_roots: int, max_subgraphs: int):
        start, cnt, serial = C.c_uint64(), C.c_uint32(), C.c_uint64()
        check(self._lib.sg_next_roots(self._h, num_roots, max_subgraphs, C.byref(start),
                                      C.byref(cnt), C.byref(serial)))
        return start.value, cnt.value, serial.value

    # ------------------------------------------------------------------- ppr
    def set_ppr(self, targets, length, neigh, score):
        t = np.ascontiguousarray(targets, dtype=np.uint32)
        ln = np.ascontiguousarray(length, dtype=np.uint32)
        nb = np.ascontiguousarray(neigh, dtype=np.uint32)
        sc = np.ascontiguousarray(score, dtype=np.float32)
        assert nb.ndim == 2 and nb.shape == sc.shape and nb.shape[0] == t.size == ln.size
        check(self._lib.sg_set_ppr(self._h, t.ctypes.data, t.size, ln.ctypes.data, nb.ctypes.data,
                                   sc.ctypes.data, nb.shape[1]))

    def load_ppr_bin(self, fname_neighs, fname_scores, k, alpha, epsilon):
        check(self._lib.sg_load_ppr_bin(self._h, fname_neighs.encode(), fname_scores.encode(), k,
                                        alpha, epsilon))

    def save_ppr_bin(self, fname_neighs, fname_scores, k, alpha, epsilon):
        check(self._lib.sg_save_ppr_bin(self._h, fname_neighs.encode(), fname_scores.encode(), k,
                                        alpha, epsilon))

    def drop_full_graph_info(self):
        check(self._lib.sg_drop_full_graph_info(self._h))

    def set_profiling(self, enable: bool):
        check(self._lib.sg_set_profiling(self._h, int(bool(enable))))

    def set_caps(self, cap_subg_nodes=0, cap_subg_edges=0):
        check(self._lib.sg_set_caps(self._h, cap_subg_nodes, cap_subg_edges))

    def get_caps(self, cfg: SamplerConfig):
        c = cfg.to_c()
        a, b = C.c_uint32(), C.c_uint32()
        check(self._lib.sg_get_caps(self._h, C.byref(c), C.byref(a), C.byref(b)))
        return a.value, b.value

    # -------------------------------------------------------------- sampling
    def _alloc(self, cfg: SamplerConfig, P: int, cap_nodes: int, cap_edges: int):
        dev = self.device
        i32 = dict(dtype=torch.int32, device=dev)
        if cap_nodes >= (1 << 32) or cap_edges >= (1 << 32):
            raise CapacityError(_lib.SG_ERR_CAPACITY, f"output buffers of {cap_nodes} nodes / {cap_edges} edges for {P} "
                                f"subgraphs exceed the uint32 batch index space; use smaller sampler calls")
        b = dict(node=torch.empty(cap_nodes, **i32), indptr=torch.empty(cap_nodes + 1, **i32),
                 indices=torch.empty(cap_edges, **i32), edge_id=torch.empty(cap_edges, **i32),
                 target=torch.empty(max(1, P * cfg.num_roots), **i32),
                 subg_node_off=torch.empty(P + 1, **i32), subg_edge_off=torch.empty(P + 1, **i32),
                 ppr=torch.empty(cap_nodes, dtype=torch.float32, device=dev),
                 hop=torch.empty(cap_nodes, **i32) if "hops" in cfg.aug else None,
                 drnl=torch.empty(cap_nodes, **i32) if "drnls" in cfg.aug else None)
        out = SgBatchOut(b["node"].data_ptr(), b["indptr"].data_ptr(), b["indices"].data_ptr(),
                         b["edge_id"].data_ptr(), b["target"].data_ptr(),
                         b["subg_node_off"].data_ptr(), b["subg_edge_off"].data_ptr(),
                         b["hop"].data_ptr() if b["hop"] is not None else None,
                         b["ppr"].data_ptr(),
                         b["drnl"].data_ptr() if b["drnl"] is not None else None,
                         cap_nodes, cap_edges)
        return b, out

    @staticmethod
    def _cfg_key(cfg: SamplerConfig):
        return (cfg.method, cfg.num_roots, cfg.depth, cfg.budget, cfg.k, cfg.add_self_edge)

    def _out_caps(self, cfg, P, cap_edges_out, cap_nodes_out):
        capn, cape = self.get_caps(cfg)
        if cap_nodes_out is None:
            # worst case P * capn when that is small; otherwise (hub-heavy graphs, grown caps) the running
            # high-water mark with head room -- finish() re-runs with the exact size on overflow
            cap_nodes_out = P * capn
            if cap_nodes_out > (1 << 24):
                cap_nodes_out = min(cap_nodes_out, max(1 << 24, P * (self._hwm.get(self._cfg_key(cfg), (0, 0))[1] * 3 // 2 + 64)))
        cap_nodes_out = max(1, cap_nodes_out)
        if cap_edges_out is None:
            per = max(self._hwm.get(self._cfg_key(cfg), (0, 0))[0] * 3 // 2, min(cape, 8 * capn), 64)
            cap_edges_out = max(1, P * per)
        cap_edges_out = min(cap_edges_out, max(1, P * cape))
        return cap_nodes_out, cap_edges_out

    def _launch(self, pend: _Pending, cap_edges_out=None, cap_nodes_out=None):
        """Allocate the outputs and enqueue the call.  ``pend.sizes`` (optional): the call fills SEVERAL consecutive
        batches (sg_sample_multi); the caps are then lists (or None) with one entry per batch."""
        cfg, P = pend.cfg, pend.P
        sizes = getattr(pend, "sizes", None)
        c = cfg.to_c()
        stream = torch._C._cuda_getCurrentRawStream(self.device.index)
        roots_ptr = pend.roots_dev.data_ptr() if pend.roots_dev is not None else None
        if not sizes:
            cn, ce = self._out_caps(cfg, P, cap_edges_out, cap_nodes_out)
            pend.bufs, pend.out = self._alloc(cfg, P, cn, ce)
            check(self._lib.sg_sample(self._h, C.byref(c), pend.root_start, P, pend.serial, roots_ptr,
                                      C.byref(pend.out), stream))
            return
        S = len(sizes)
        outs = (SgBatchOut * S)()
        pend.bufs = []
        for i, Pi in enumerate(sizes):
            cn, ce = self._out_caps(cfg, Pi, cap_edges_out[i] if cap_edges_out else None, cap_nodes_out[i] if cap_nodes_out else None)
            b, o = self._alloc(cfg, Pi, cn, ce)
            pend.bufs.append(b)
            outs[i] = o
        pend.out = outs
        check(self._lib.sg_sample_multi(self._h, C.byref(c), pend.root_start, S, (C.c_uint32 * S)(*sizes), pend.serial, roots_ptr,
                                        outs, stream))

    def sample_async(self, cfg: SamplerConfig, max_subgraphs: int = 0, *, roots=None,
                     serial_base: Optional[int] = None):
        """Enqueue one sampler call on the current torch stream.  Either the next
        ``max_subgraphs`` root groups of the shuffled target list (sequential
        cursor, ParallelSampler.cpp:456-468) or an explicit ``roots`` array."""
        if self._pending is not None:
            raise RuntimeError("a sample is already in flight; call finish() first")
        pend = _Pending()
        pend.cfg = cfg
        pend.roots_dev = None
        pend.sizes = None
        if roots is not None:
            r = torch.as_tensor(np.ascontiguousarray(np.asarray(roots).reshape(-1), dtype=np.uint32).view(np.int32))
            assert r.numel() % cfg.num_roots == 0
            pend.roots_dev = r.to(self.device)
            pend.P = r.numel() // cfg.num_roots
            pend.root_start = 0
            pend.serial = 0 if serial_base is None else serial_base
        else:
            pend.root_start, pend.P, pend.serial = self.next_roots(cfg.num_roots, max_subgraphs)
            if serial_base is not None:
                pend.serial = serial_base
        with torch.cuda.device(self.device):
            self._launch(pend)
        self._pending = pend
        return pend

    def sample_multi_async(self, cfg: SamplerConfig, sizes, *, roots=None, serial_base: Optional[int] = None):
        """ONE sampler call for several consecutive batches (sg_sample_multi): batch i takes the next ``sizes[i]`` root
        groups of the shuffled target list behind batch i - 1 (or of ``roots``).  Subgraph for subgraph the result equals
        ``len(sizes)`` separate ``sample_async`` calls -- the draws are keyed on the subgraph's serial number -- but the
        pipeline's four dependent launches are paid once.  The root cursor must not wrap inside the call: the sizes
        have to fit what is left of the target list (a trainer's epoch plan guarantees that)."""
        if self._pending is not None:
            raise RuntimeError("a sample is already in flight; call finish() first")
        sizes = [int(x) for x in sizes]
        if not (1 <= len(sizes) <= _lib.MAX_BATCHES_PER_CALL) or min(sizes) < 1:
            raise ValueError(f"1..{_lib.MAX_BATCHES_PER_CALL} non-empty batches per call, got {sizes}")
        pend = _Pending()
        pend.cfg, pend.roots_dev, pend.sizes, pend.P = cfg, None, sizes, sum(sizes)
        if roots is not None:
            r = torch.as_tensor(np.ascontiguousarray(np.asarray(roots).reshape(-1), dtype=np.uint32).view(np.int32))
            if r.numel() != pend.P * cfg.num_roots:
                raise ValueError(f"{r.numel()} roots for batches of {sizes} x {cfg.num_roots}")
            pend.roots_dev = r.to(self.device)
            pend.root_start = 0
            pend.serial = 0 if serial_base is None else serial_base
        else:
            left = (self.num_nodes_target() - self.get_idx_root()) // cfg.num_roots
            if pend.P > left:
                raise ValueError(f"batches of {sizes} subgraphs do not fit the {left} root groups left before the cursor wraps")
            pend.root_start, got, pend.serial = self.next_roots(cfg.num_roots, pend.P)
            assert got == pend.P
            if serial_base is not None:
                pend.serial = serial_base
        with torch.cuda.device(self.device):
            self._launch(pend)
        self._pending = pend
        return pend

    def _wrap(self, cfg, b, cnt, P, call=None) -> DeviceBatch:
        n, e = int(cnt.n_tot), int(cnt.e_tot)
        counts = dict(n_tot=n, e_tot=e, max_subg_nodes=cnt.max_subg_nodes,
                      max_subg_edges=cnt.max_subg_edges, slots_scanned=int(cnt.slots_scanned),
                      frontier_reads=int(cnt.frontier_reads), frontier_nodes=int(cnt.frontier_nodes),
                      sample_kernel_ms=float(cnt.sample_kernel_ms),
                      relocate_kernel_ms=float(cnt.relocate_kernel_ms))
        if call is not None:           # (batch `index` of a `batches`-batch call: the call's kernel times sit on index 0)
            counts.update(call_id=call[0], call_index=call[1], call_batches=call[2])
        return DeviceBatch(
            node=b["node"][:n], indptr=b["indptr"][:n + 1], indices=b["indices"][:e],
            edge_id=b["edge_id"][:e], target=b["target"][:P * cfg.num_roots],
            subg_node_off=b["subg_node_off"], subg_edge_off=b["subg_edge_off"], ppr=b["ppr"][:n],
            hop=b["hop"][:n] if b["hop"] is not None else None,
            drnl=b["drnl"][:n] if b["drnl"] is not None else None,
            num_subgraphs=P, num_roots=cfg.num_roots, counts=counts)

    def finish_multi(self, on_retry=None) -> List[DeviceBatch]:
        """Wait for the in-flight multi-batch call; returns its batches in order.  A capacity overflow of any batch grows
        the caps and re-runs the WHOLE call with the same roots / serials."""
        pend = self._pending
        if pend is None or not getattr(pend, "sizes", None):
            raise RuntimeError("no multi-batch sample in flight")
        S = len(pend.sizes)
        cnts = (SgBatchCounts * S)()
        for _ in range(12):
            rc = self._lib.sg_sample_finish_multi(self._h, S, cnts)
            if rc == _lib.SG_OK:
                break
            if rc != _lib.SG_ERR_CAPACITY:
                self._pending = None
                check(rc)
            ov = 0
            for c_ in cnts:
                ov |= c_.overflow
            if ov & 1:
                self.set_caps(cap_subg_nodes=min(self.num_nodes(), max(2 * max(c_.max_subg_nodes for c_ in cnts), 1024)))
            if ov & 2:
                m = max(c_.max_subg_edges for c_ in cnts)
                self.set_caps(cap_subg_edges=m + m // 4 + 64)
            cap_e = cap_n = None
            if (ov & 8) and not (ov & 3):
                cap_e = [int(c_.e_tot) + int(c_.e_tot) // 8 + 64 if (c_.overflow & 8) else None for c_ in cnts]
            if (ov & 4) and not (ov & 3):
                cap_n = [int(c_.n_tot) + int(c_.n_tot) // 8 + 64 if (c_.overflow & 4) else None for c_ in cnts]
            if on_retry is not None:
                on_retry()
            with torch.cuda.device(self.device):
                self._launch(pend, cap_e, cap_n)
        else:
            self._pending = None
            raise CapacityError(_lib.SG_ERR_CAPACITY, "sampler capacity did not converge")
        self._pending = None
        k = self._cfg_key(pend.cfg)
        self._call_serial = getattr(self, "_call_serial", 0) + 1
        out = []
        for i, (Pi, b, cnt) in enumerate(zip(pend.sizes, pend.bufs, cnts)):
            he, hn = self._hwm.get(k, (0, 0))
            self._hwm[k] = (max(he, -(-int(cnt.e_tot) // Pi)), max(hn, -(-int(cnt.n_tot) // Pi)))
            out.append(self._wrap(pend.cfg, b, cnt, Pi, call=(self._call_serial, i, S)))
        return out

    def sample_multi(self, cfg: SamplerConfig, sizes, *, roots=None, serial_base: Optional[int] = None) -> List[DeviceBatch]:
        self.sample_multi_async(cfg, sizes, roots=roots, serial_base=serial_base)
        return self.finish_multi()

    def finish(self, on_retry=None) -> DeviceBatch:
        """Wait for the in-flight call and wrap the outputs.  Capacity overflows
        are handled here by growing and re-running the same roots / serials;
        ``on_retry()`` is called before such a re-run allocates and launches again
        (a caller that prefetches on a side stream orders it behind the consumers
        of the blocks the allocator may hand back)."""
        pend = self._pending
        if pend is None:
            raise RuntimeError("no sample in flight")
        if getattr(pend, "sizes", None):
            raise RuntimeError("a multi-batch call is in flight: finish_multi()")
        cnt = SgBatchCounts()
        for _ in range(12):
            rc = self._lib.sg_sample_finish(self._h, C.byref(cnt))
            if rc == _lib.SG_OK:
                break
            if rc != _lib.SG_ERR_CAPACITY:
                self._pending = None
                check(rc)
            ov = cnt.overflow
            # (ov & 16 -- the scan's round-record pool -- is repaired inside the library: the re-run below finds it doubled)
            if ov & 1:
                self.set_caps(cap_subg_nodes=min(self.num_nodes(), max(2 * cnt.max_subg_nodes, 1024)))
            if ov & 2:
                self.set_caps(cap_subg_edges=cnt.max_subg_edges + cnt.max_subg_edges // 4 + 64)
            cap_e = cap_n = None
            if (ov & 8) and not (ov & 3):
                cap_e = int(cnt.e_tot) + int(cnt.e_tot) // 8 + 64
            if (ov & 4) and not (ov & 3):
                cap_n = int(cnt.n_tot) + int(cnt.n_tot) // 8 + 64
            if on_retry is not None:
                on_retry()
            with torch.cuda.device(self.device):
                self._launch(pend, cap_e, cap_n)
        else:
            self._pending = None
            raise CapacityError(_lib.SG_ERR_CAPACITY, "sampler capacity did not converge")
        self._pending = None
        n, e, P = int(cnt.n_tot), int(cnt.e_tot), pend.P
        if P:
            k = self._cfg_key(pend.cfg)
            he, hn = self._hwm.get(k, (0, 0))
            self._hwm[k] = (max(he, -(-e // P)), max(hn, -(-n // P)))
        return self._wrap(pend.cfg, pend.bufs, cnt, P)

    def sample(self, cfg: SamplerConfig, max_subgraphs: int = 0, *, roots=None,
               serial_base: Optional[int] = None) -> DeviceBatch:
        self.sample_async(cfg, max_subgraphs, roots=roots, serial_base=serial_base)
        return self.finish()


class SubgraphCache:
    """Device-resident record -> reuse cache of sampled subgraphs keyed by root id: the
    reference's CachedSubgraph + PoolSubgraph.collate for deterministic samplers
    (shaDow/minibatch.py:21-91, :403-426) without leaving HBM."""

    def __init__(self, num_nodes: int, device: torch.device):
        self._lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("SubgraphCache needs a ROCm device (torch device type 'cuda')")
        if self.device.index is None:            # 'cuda' means the CURRENT device, as for HipSampler
            self.device = torch.device("cuda", torch.cuda.current_device())
        h = C.c_void_p()
        check(self._lib.sg_cache_create(num_nodes, self.device.index, C.byref(h)))
        self._h = h
        self._pend = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sg_cache_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        check(self._lib.sg_cache_clear(self._h))

    def stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self._lib.sg_cache_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(num_recorded=a.value, nodes=b.value, edges=c.value)

    def is_empty(self):
        return self.stats()["num_recorded"] == 0

    def record(self, batch: DeviceBatch):
        """File every (single-root) subgraph of a sampled batch under its root id."""
        assert batch.num_roots == 1, "the cache holds node-task subgraphs (minibatch.py:410)"
        if batch.node.device != self.device:
            raise ValueError(f"batch lives on {batch.node.device}, the cache on {self.device}")
        out = SgBatchOut(batch.node.data_ptr(), batch.indptr.data_ptr(), batch.indices.data_ptr(),
                         batch.edge_id.data_ptr(), batch.target.data_ptr(), batch.subg_node_off.data_ptr(),
                         batch.subg_edge_off.data_ptr(), batch.hop.data_ptr() if batch.hop is not None else None,
                         batch.ppr.data_ptr(), None, batch.num_nodes, batch.num_edges)
        stream = torch._C._cuda_getCurrentRawStream(self.device.index)
        with torch.cuda.device(self.device):
            check(self._lib.sg_cache_record(self._h, C.byref(out), batch.num_subgraphs, batch.num_nodes,
                                            batch.num_edges, stream))

    def _alloc(self, P, cap_nodes, cap_edges, want_hop):
        i32 = dict(dtype=torch.int32, device=self.device)
        b = dict(node=torch.empty(cap_nodes, **i32), indptr=torch.empty(cap_nodes + 1, **i32),
                 indices=torch.empty(cap_edges, **i32), edge_id=torch.empty(cap_edges, **i32),
                 target=torch.empty(max(1, P), **i32), subg_node_off=torch.empty(P + 1, **i32),
                 subg_edge_off=torch.empty(P + 1, **i32),
                 ppr=torch.empty(cap_nodes, dtype=torch.float32, device=self.device),
                 hop=torch.empty(cap_nodes, **i32) if want_hop else None)
        out = SgBatchOut(b["node"].data_ptr(), b["indptr"].data_ptr(), b["indices"].data_ptr(),
                         b["edge_id"].data_ptr(), b["target"].data_ptr(), b["subg_node_off"].data_ptr(),
                         b["subg_edge_off"].data_ptr(), b["hop"].data_ptr() if want_hop else None,
                         b["ppr"].data_ptr(), None, cap_nodes, cap_edges)
        return b, out

    def collate_async(self, roots, cap_nodes: int, cap_edges: int, want_hop: bool = False):
        """Start rebuilding the block-diagonal batch of ``roots`` (device int32 tensor or array)."""
        if self._pend is not None:
            raise RuntimeError("a collate is already in flight; call finish() first")
        if not (isinstance(roots, torch.Tensor) and roots.is_cuda):
            roots = torch.as_tensor(np.ascontiguousarray(np.asarray(roots).reshape(-1), dtype=np.uint32).view(np.int32)).to(self.device)
        roots = roots.contiguous()
        P = int(roots.numel())
        bufs, out = self._alloc(P, max(1, cap_nodes), max(1, cap_edges), want_hop)
        stream = torch._C._cuda_getCurrentRawStream(self.device.index)
        with torch.cuda.device(self.device):
            check(self._lib.sg_cache_collate(self._h, roots.data_ptr(), P, C.byref(out), stream))
        self._pend = (roots, bufs, out, P, want_hop)

    def finish(self, on_retry=None) -> DeviceBatch:
        if self._pend is None:
            raise RuntimeError("no collate in flight")
        roots, b, out, P, want_hop = self._pend
        cnt = SgBatchCounts()
        rc = self._lib.sg_cache_collate_finish(self._h, C.byref(cnt))
        self._pend = None
        if rc == _lib.SG_ERR_CAPACITY:          # sizes are known now: run again with exact buffers
            if on_retry is not None:
                on_retry()
            self.collate_async(roots, int(cnt.n_tot), int(cnt.e_tot), want_hop)
            return self.finish()
        check(rc)
        n, e = int(cnt.n_tot), int(cnt.e_tot)
        return DeviceBatch(node=b["node"][:n], indptr=b["indptr"][:n + 1], indices=b["indices"][:e],
                           edge_id=b["edge_id"][:e], target=b["target"][:P], subg_node_off=b["subg_node_off"],
                           subg_edge_off=b["subg_edge_off"], ppr=b["ppr"][:n],
                           hop=b["hop"][:n] if want_hop else None, drnl=None, num_subgraphs=P, num_roots=1,
                           counts=dict(n_tot=n, e_tot=e, max_subg_nodes=cnt.max_subg_nodes,
                                       max_subg_edges=cnt.max_subg_edges, slots_scanned=0, frontier_reads=0,
                                       frontier_nodes=0, sample_kernel_ms=0.0, relocate_kernel_ms=0.0))

    def collate(self, roots, cap_nodes: int, cap_edges: int, want_hop: bool = False) -> DeviceBatch:
        self.collate_async(roots, cap_nodes, cap_edges, want_hop)
        return self.finish()


# ===========================================================================
# Reference-compatible surface (pybind11 module `ParallelSampler`)
# ===========================================================================
class SubgraphStructVec:
    """Mirror of the reference's SubgraphStructVec (Graph.h:59-97, bindings
    ParallelSampler.cpp:735-745): struct-of-vectors over one sampler call.  Each
    getter returns a list with ``num_sampler_per_batch`` entries (one sequence
    of ints per subgraph; entries past get_num_valid_subg() are empty), exactly
    like the C++ vectors the frontend slices with ``[:clip]``."""

    _FIELDS = ("indptr", "indices", "data", "node", "edge_index", "target", "hop", "ppr", "drnl")

    def __init__(self, num_subgraphs: int):
        self._cap = num_subgraphs
        self._valid = 0
        self._v = {f: [np.zeros(0, dtype=np.float32 if f in ("data", "ppr") else np.int64)
                       for _ in range(num_subgraphs)] for f in self._FIELDS}

    def _fill_from_batch(self, batch: DeviceBatch, aug):
        subs = batch.split_host()
        self._valid = len(subs)
        for p, s in enumerate(subs):
            self._v["indptr"][p] = s["indptr"]
            self._v["indices"][p] = s["indices"]
            self._v["data"][p] = np.ones(s["indices"].size, dtype=np.float32)   # .cpp:411,423
            self._v["node"][p] = s["node"]
            # NodeType(-1) surfaces as 4294967295 through pybind
            self._v["edge_index"][p] = s["edge_index"]
            self._v["target"][p] = s["target"]
            self._v["ppr"][p] = s["ppr"]
            if "hop" in s:
                self._v["hop"][p] = s["hop"]
            if "drnl" in s:
                self._v["drnl"][p] = s["drnl"]

    def _fill_targets_only(self, roots: np.ndarray, num_roots: int):
        # ParallelSampler::dummy_sampler, .cpp:653-659: only origNodeID is filled
        P = roots.size // num_roots
        self._valid = P
        for p in range(P):
            self._v["node"][p] = roots[p * num_roots:(p + 1) * num_roots].astype(np.int64)

    def get_num_valid_subg(self):
        return self._valid

    def get_subgraph_indptr(self):
        return self._v["indptr"]

    def get_subgraph_indices(self):
        return self._v["indices"]

    def get_subgraph_data(self):
        return self._v["data"]

    def get_subgraph_node(self):
        return self._v["node"]

    def get_subgraph_edge_index(self):
        return self._v["edge_index"]

    def get_subgraph_target(self):
        return self._v["target"]

    def get_subgraph_hop(self):
        return self._v["hop"]

    def get_subgraph_ppr(self):
        return self._v["ppr"]

    def get_subgraph_drnl(self):
        return self._v["drnl"]


class ParallelSampler:
    """Mirror of the reference's ``ParallelSampler`` class (constructor
    ParallelSampler.h:27-69, bindings ParallelSampler.cpp:708-734).

    Positional constructor, same order as the pybind signature:
    (indptr, indices, data, num_sampler_per_batch, max_num_threads, fix_target,
     sequential_traversal, edge_reweighted, num_subgraphs_ensemble,
     path_indptr, path_indices, path_data, seed).
    ``data`` / ``edge_reweighted`` / ``path_data`` / ``max_num_threads`` are
    accepted and ignored exactly as far as the reference ignores them
    (ParallelSampler.h:48; threads are a CPU notion)."""

    def __init__(self, indptr, indices, data, num_sampler_per_batch, max_num_threads,
                 fix_target, sequential_traversal, edge_reweighted=(), num_subgraphs_ensemble=1,
                 path_indptr="", path_indices="", path_data="", seed=-1, device=None):
        if not sequential_traversal:
            # the reference's random-root branch (.cpp:469-478) is dead code
            # (asserted off in frontend/samplers_ensemble.py:93)
            raise NotImplementedError("only sequential root traversal is supported")
        self.num_sampler_per_batch = int(num_sampler_per_batch)
        self.fix_target = bool(fix_target)
        self.sequential_traversal = True
        self.num_subgraphs_ensemble = int(num_subgraphs_ensemble)
        self._hip = HipSampler(indptr, indices, device=device, seed=seed,
                               path_indptr=path_indptr, path_indices=path_indices)
        self._targets = np.zeros(0, dtype=np.uint32)
        self._seed = seed

    # -- utility functions (.cpp:45-53)
    def num_nodes(self):
        return self._hip.num_nodes()

    def num_edges(self):
        return self._hip.num_edges()

    def num_nodes_target(self):
        return self._hip.num_nodes_target()

    def get_idx_root(self):
        return self._hip.get_idx_root()

    def is_seq_root_traversal(self):
        return self.sequential_traversal

    def shuffle_targets(self, targets_pre_shuffled):
        """ParallelSampler::shuffle_targets (.cpp:36-43): empty argument =>
        shuffle the stored list in place, otherwise adopt the given order."""
        t = np.asarray(targets_pre_shuffled, dtype=np.uint32).reshape(-1)
        if t.size == 0:
            rng = np.random.default_rng(None if self._seed is None or self._seed < 0 else self._seed)
            t = self._targets.copy()
            rng.shuffle(t)
        else:
            assert self._targets.size == 0 or self._targets.size == t.size
        self._targets = t
        self._hip.shuffle_targets(t)

    def drop_full_graph_info(self):
        self._hip.drop_full_graph_info()

    def preproc_ppr_approximate(self, preproc_target, k, alpha, epsilon, fname_neighs, fname_scores):
        """ParallelSampler::preproc_ppr_approximate (.cpp:237-344): use the cache
        files when they match, else compute the table and write the files."""
        if fname_neighs and fname_scores:
            try:
                self._hip.load_ppr_bin(fname_neighs, fname_scores, int(k), float(alpha), float(epsilon))
                return
            except _lib.ShadowHipError as e:
                if e.code != _lib.SG_ERR_IO:
                    raise
        from .ppr import ppr_approximate_device   # HIP push kernel
        t = np.asarray(preproc_target, dtype=np.uint32).reshape(-1)
        ln, nb, sc = ppr_approximate_device(self._hip, t, int(k), float(alpha), float(epsilon))
        self._hip.set_ppr(t, ln, nb, sc)
        if fname_neighs and fname_scores:
            self._hip.save_ppr_bin(fname_neighs, fname_scores, int(k), float(alpha), float(epsilon))

    def parallel_sampler_ensemble(self, configs_samplers: List[Dict[str, str]],
                                  configs_aug: List[set]) -> List[SubgraphStructVec]:
        """ParallelSampler::parallel_sampler_ensemble (.cpp:662-704): one root
        window shared by all ensemble branches."""
        assert len(configs_samplers) == self.num_subgraphs_ensemble
        cfgs = [SamplerConfig.from_cpp_dict(c, a) for c, a in zip(configs_samplers, configs_aug)]
        num_roots = cfgs[0].num_roots
        assert all(c.num_roots == num_roots for c in cfgs)          # .cpp:668-671
        start, P, serial = self._hip.next_roots(num_roots, self.num_sampler_per_batch)
        ret = []
        for i, cfg in enumerate(cfgs):
            vec = SubgraphStructVec(self.num_sampler_per_batch)
            if cfg.return_target_only:
                vec._fill_targets_only(self._targets[start:start + P * num_roots], num_roots)
            else:
                pend = _Pending()
                pend.cfg, pend.P, pend.root_start, pend.roots_dev, pend.sizes = cfg, P, start, None, None
                # every branch draws from its own serial range
                pend.serial = serial + i * (1 << 40)
                with torch.cuda.device(self._hip.device):
                    self._hip._launch(pend)
                self._hip._pending = pend
                vec._fill_from_batch(self._hip.finish(), cfg.aug)
            ret.append(vec)
        return ret
