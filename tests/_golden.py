"""Helpers to read the committed golden fixtures (tests/golden/*.npz).

The fixtures were produced by oracle/gen_golden.py from the reference's own
C++ sampler; they are data only."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAMPLER_FIXTURES = ["path6", "rand300", "selfloop500", "directed400"]
FIELDS = ["indptr", "indices", "node", "edge_index", "target", "hop", "ppr", "drnl"]


class Fixture:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, f"sampler_{name}.npz"))
        self.indptr = self.z["indptr"]
        self.indices = self.z["indices"]
        self.cases = json.loads(bytes(self.z["cases"]).decode())

    def roots(self, ci):
        return self.z[f"c{ci}_roots"]

    def ref_subgraphs(self, ci):
        """list of dict(field -> array) as the reference returned them"""
        z = self.z
        P = z[f"c{ci}_node_off"].size - 1
        out = []
        for p in range(P):
            d = {}
            for f in FIELDS:
                off = z[f"c{ci}_{f}_off"]
                a = z[f"c{ci}_{f}"][off[p]:off[p + 1]]
                d[f] = a.astype(np.float32) if f == "ppr" else a.astype(np.int64)
            out.append(d)
        return out

    def ppr_table(self, ci):
        """(targets, len, neigh, score) decoded from the reference's cache files"""
        z = self.z
        if f"c{ci}_ppr_len" not in z:
            return None
        return self.roots(ci), z[f"c{ci}_ppr_len"], z[f"c{ci}_ppr_neigh"], z[f"c{ci}_ppr_score"]


def has_self_loops(indptr, indices):
    N = indptr.size - 1
    rows = np.repeat(np.arange(N), np.diff(indptr.astype(np.int64)))
    return bool(np.any(rows == indices))


def touches_end_of_indices(indptr, node_ids):
    """True when a node's row ends exactly at nnz: the reference's over-read
    (ParallelSampler.cpp:401-405) then reads past the vector (undefined)."""
    nnz = int(indptr[-1])
    return bool(np.any(indptr[np.asarray(node_ids, dtype=np.int64) + 1] == nnz))


def sampler_kwargs(case):
    cfg = dict(case["cfg"])
    kw = dict(method=cfg["method"], num_roots=int(cfg["num_roots"]),
              add_self_edge=bool(cfg.get("add_self_edge", False)),
              include_target_conn=bool(cfg.get("include_target_conn", False)),
              aug=tuple(case["aug"]))
    if cfg["method"] == "khop":
        kw.update(depth=int(cfg["depth"]), budget=int(cfg["budget"]))
    if cfg["method"] == "ppr":
        kw.update(k=int(cfg["k"]), threshold=float(cfg["threshold"]))
    return kw
