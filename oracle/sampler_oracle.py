"""ctypes front-end of oracle/sampler_oracle.c.

TEST INFRASTRUCTURE ONLY -- the product path (shadow_gnn_amd/) never imports
this module.  Users: tests/, __graft_entry__.smoke(), bench.py cpu_baseline.
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB_PATH = os.path.join(_BUILD, "libsampler_oracle.so")
_SRC = os.path.join(_HERE, "sampler_oracle.c")

METHOD = {"khop": 0, "ppr": 1, "nodeIID": 2}


def build(force: bool = False) -> str:
    """gcc the C restatement into oracle/_build/ (no FMA contraction: the PPR
    push must round exactly like the reference's scalar fp32 code)."""
    os.makedirs(_BUILD, exist_ok=True)
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(_SRC)):
        return _LIB_PATH
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-std=c11",
           "-Wall", "-Wextra", _SRC, "-o", _LIB_PATH, "-lm"]
    subprocess.check_call(cmd)
    return _LIB_PATH


class _Config(C.Structure):
    _fields_ = [
        ("method", C.c_int32), ("num_roots", C.c_int32), ("depth", C.c_int32),
        ("budget", C.c_int32), ("k", C.c_int32), ("threshold", C.c_float),
        ("add_self_edge", C.c_int32), ("include_target_conn", C.c_int32),
        ("compat_overread", C.c_int32), ("aug_hops", C.c_int32), ("aug_pprs", C.c_int32),
        ("aug_drnls", C.c_int32),
    ]


class _PprTable(C.Structure):
    _fields_ = [
        ("row_of_node", C.POINTER(C.c_int32)), ("len", C.POINTER(C.c_uint32)),
        ("neigh", C.POINTER(C.c_uint32)), ("score", C.POINTER(C.c_float)),
        ("stride", C.c_uint32),
    ]


class _Batch(C.Structure):
    _fields_ = [
        ("num_subg", C.c_uint32), ("n_tot", C.c_uint64), ("e_tot", C.c_uint64),
        ("node", C.POINTER(C.c_uint32)), ("indptr", C.POINTER(C.c_uint32)),
        ("indices", C.POINTER(C.c_uint32)), ("edge_id", C.POINTER(C.c_uint32)),
        ("target", C.POINTER(C.c_uint32)), ("subg_nodes", C.POINTER(C.c_uint32)),
        ("subg_edges", C.POINTER(C.c_uint32)), ("hop", C.POINTER(C.c_uint32)),
        ("drnl", C.POINTER(C.c_uint32)), ("ppr", C.POINTER(C.c_float)),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_sample_batch.restype = C.c_int
        _lib.orc_sample_batch.argtypes = [
            C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(_PprTable), C.c_void_p,
            C.c_uint32, C.POINTER(_Config), C.c_uint64, C.c_uint64, C.c_int, C.POINTER(_Batch)]
        _lib.orc_free_batch.argtypes = [C.POINTER(_Batch)]
        _lib.orc_ppr_approximate.restype = C.c_int
        _lib.orc_ppr_approximate.argtypes = [
            C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_float,
            C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_draw_offset.restype = C.c_uint32
        _lib.orc_draw_offset.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32]
    return _lib


@dataclass
class PprTable:
    """Compact top-k PPR table (one row per preprocessed target)."""
    row_of_node: np.ndarray  # int32 [N]
    len: np.ndarray          # uint32 [R]
    neigh: np.ndarray        # uint32 [R, stride]
    score: np.ndarray        # float32 [R, stride]

    @property
    def stride(self):
        return int(self.neigh.shape[1])


@dataclass
class Batch:
    """Block-diagonal batch of sampled subgraphs (numpy, host)."""
    node: np.ndarray
    indptr: np.ndarray
    indices: np.ndarray
    edge_id: np.ndarray
    target: np.ndarray
    subg_nodes: np.ndarray
    subg_edges: np.ndarray
    ppr: np.ndarray
    hop: Optional[np.ndarray] = None
    drnl: Optional[np.ndarray] = None

    def split(self):
        """Per-subgraph local views, in the reference getter convention
        (indptr / indices / node / edge_index / target [/hop/ppr/drnl])."""
        out = []
        no = np.concatenate([[0], np.cumsum(self.subg_nodes, dtype=np.int64)])
        eo = np.concatenate([[0], np.cumsum(self.subg_edges, dtype=np.int64)])
        R = self.target.size // max(1, self.subg_nodes.size)
        for p in range(self.subg_nodes.size):
            n0, n1, e0, e1 = no[p], no[p + 1], eo[p], eo[p + 1]
            d = dict(
                indptr=(self.indptr[n0:n1 + 1].astype(np.int64) - e0),
                indices=(self.indices[e0:e1].astype(np.int64) - n0),
                node=self.node[n0:n1].astype(np.int64),
                edge_index=self.edge_id[e0:e1].astype(np.int64),
                target=(self.target[p * R:(p + 1) * R].astype(np.int64) - n0),
                ppr=self.ppr[n0:n1],
            )
            if self.hop is not None:
                d["hop"] = self.hop[n0:n1].astype(np.int64)
            if self.drnl is not None:
                d["drnl"] = self.drnl[n0:n1].astype(np.int64)
            out.append(d)
        return out


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _np_from(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True)


def sample_batch(indptr, indices, roots, *, method="khop", num_roots=1, depth=2, budget=-1, k=0,
                 threshold=0.0, add_self_edge=False, include_target_conn=False,
                 compat_overread=False, aug=(), ppr: Optional[PprTable] = None, seed=0,
                 serial_base=0, num_threads=1) -> Batch:
    indptr = _u32(indptr)
    indices = _u32(indices)
    roots = _u32(roots).reshape(-1)
    assert roots.size % num_roots == 0
    P = roots.size // num_roots
    cfg = _Config(METHOD[method], num_roots, depth, budget, k, threshold, int(add_self_edge),
                  int(include_target_conn), int(compat_overread), int("hops" in aug),
                  int("pprs" in aug), int("drnls" in aug))
    tab = None
    keep = []
    if ppr is not None:
        ron = np.ascontiguousarray(ppr.row_of_node, dtype=np.int32)
        ln = _u32(ppr.len)
        nb = _u32(ppr.neigh)
        sc = np.ascontiguousarray(ppr.score, dtype=np.float32)
        keep = [ron, ln, nb, sc]
        tab = _PprTable(ron.ctypes.data_as(C.POINTER(C.c_int32)),
                        ln.ctypes.data_as(C.POINTER(C.c_uint32)),
                        nb.ctypes.data_as(C.POINTER(C.c_uint32)),
                        sc.ctypes.data_as(C.POINTER(C.c_float)), ppr.stride)
    b = _Batch()
    rc = lib().orc_sample_batch(indptr.ctypes.data, indices.ctypes.data, indptr.size - 1,
                                indices.size, C.byref(tab) if tab is not None else None,
                                roots.ctypes.data, P, C.byref(cfg), seed, serial_base,
                                num_threads, C.byref(b))
    del keep
    if rc != 0:
        raise RuntimeError(f"orc_sample_batch failed rc={rc}")
    try:
        n, e = int(b.n_tot), int(b.e_tot)
        out = Batch(
            node=_np_from(b.node, n, np.uint32), indptr=_np_from(b.indptr, n + 1, np.uint32),
            indices=_np_from(b.indices, e, np.uint32), edge_id=_np_from(b.edge_id, e, np.uint32),
            target=_np_from(b.target, P * num_roots, np.uint32),
            subg_nodes=_np_from(b.subg_nodes, P, np.uint32),
            subg_edges=_np_from(b.subg_edges, P, np.uint32),
            ppr=_np_from(b.ppr, n, np.float32),
            hop=_np_from(b.hop, n, np.uint32) if "hops" in aug else None,
            drnl=_np_from(b.drnl, n, np.uint32) if "drnls" in aug else None,
        )
    finally:
        lib().orc_free_batch(C.byref(b))
    return out


def ppr_approximate(indptr, indices, targets, k, alpha=0.85, epsilon=1e-5, num_threads=1) -> PprTable:
    """Top-k approximate PPR rows for `targets` (reference:
    ParallelSampler::preproc_ppr_approximate).  `alpha` is the user-facing
    value (the reference flips it to 1-alpha internally, .cpp:242)."""
    indptr = _u32(indptr)
    indices = _u32(indices)
    targets = _u32(targets)
    T = targets.size
    N = indptr.size - 1
    ln = np.zeros(T, dtype=np.uint32)
    nb = np.zeros((T, k), dtype=np.uint32)
    sc = np.zeros((T, k), dtype=np.float32)
    rc = lib().orc_ppr_approximate(indptr.ctypes.data, indices.ctypes.data, N, targets.ctypes.data,
                                   T, k, alpha, epsilon, num_threads, ln.ctypes.data,
                                   nb.ctypes.data, sc.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"orc_ppr_approximate failed rc={rc}")
    row_of_node = np.full(N, -1, dtype=np.int32)
    row_of_node[targets] = np.arange(T, dtype=np.int32)
    return PprTable(row_of_node=row_of_node, len=ln, neigh=nb, score=sc)


def philox4x32_10(ctr, key):
    ctr = _u32(ctr)
    key = _u32(key)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(ctr.ctypes.data, key.ctypes.data, out.ctypes.data)
    return out


def draw_offset(seed, serial, level, v, draw, deg):
    return int(lib().orc_draw_offset(seed, serial, level, v, draw, deg))
