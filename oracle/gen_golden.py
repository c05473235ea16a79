#!/usr/bin/env python3
"""Generate the sampler golden vectors under tests/golden/ from the REFERENCE.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs
oracle/_ref, i.e. the reference's own C++ sampler compiled by
oracle/build_ref.sh from /root/reference).  The outputs are plain data
(.npz): inputs (graph CSR, roots, config) and the reference's outputs.  No
reference source text is stored.

    python oracle/gen_golden.py            # (re)writes tests/golden/sampler_*.npz

Fixture layout (one .npz per graph; cases are prefixed c<idx>_):
    indptr, indices                   the full graph (uint32 CSR)
    cases                             json list of case configs
    c<i>_roots                        [P*num_roots] roots handed to the sampler
    c<i>_{indptr,indices,node,edge_index,target,hop,ppr,drnl}  concatenated
    c<i>_{..}_off                     offsets of each subgraph in the above
    ppr_{targets,len,neigh,score}     PPR table as written by the reference's
                                      cache files (decoded), when present
tests/golden/ppr_cache_files.npz: the two cache files' raw bytes (gen_ppr_cache_files;
    `python oracle/gen_golden.py --ppr-files-only` rewrites only this one).
"""
import io
import json
import os
import struct
import sys
import tempfile

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_ref"))

FIELDS = ["indptr", "indices", "node", "edge_index", "target", "hop", "ppr", "drnl"]


def make_graph(n, avg_deg, seed, self_loops=0, directed=False):
    """Heavy-tailed random graph: one endpoint uniform, the other Pareto-weighted."""
    rng = np.random.default_rng(seed)
    m = n * avg_deg // 2
    w = rng.pareto(1.5, n) + 1.0
    w /= w.sum()
    a = rng.integers(0, n, m)
    b = rng.choice(n, size=m, p=w)
    keep = a != b
    a, b = a[keep], b[keep]
    if not directed:
        r = np.concatenate([a, b]); c = np.concatenate([b, a])
    else:
        r, c = a, b
    if self_loops:
        sl = rng.choice(n, size=self_loops, replace=False)
        r = np.concatenate([r, sl]); c = np.concatenate([c, sl])
    A = sp.csr_matrix((np.ones(r.size, dtype=np.float32), (r, c)), shape=(n, n))
    A.sum_duplicates()
    A.sort_indices()
    return A.indptr.astype(np.uint32), A.indices.astype(np.uint32)


def path_graph():
    edges = [(0, 1), (1, 2), (2, 3), (3, 4), (1, 3), (4, 5)]
    r = [a for a, b in edges] + [b for a, b in edges]
    c = [b for a, b in edges] + [a for a, b in edges]
    A = sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(6, 6))
    A.sort_indices()
    return A.indptr.astype(np.uint32), A.indices.astype(np.uint32)


def read_ppr_file(path, is_score):
    """Decode the reference's PPR cache format (ParallelSampler.cpp:105-137)."""
    raw = open(path, "rb").read()
    alpha, eps, k, cnt = struct.unpack_from("<ffiI", raw, 0)
    off = 16
    rows = []
    for _ in range(cnt):
        (ln,) = struct.unpack_from("<I", raw, off); off += 4
        rows.append(np.frombuffer(raw, dtype=np.float32 if is_score else np.uint32, count=ln, offset=off).copy())
        off += 4 * ln
    assert off == len(raw)
    return (alpha, eps, k), rows


def run_reference(indptr, indices, roots, cfg, aug, threads=1, seed=0, ppr_args=None):
    """One call of the reference sampler; returns per-field lists of arrays."""
    import ParallelSampler as ref  # oracle/_ref
    P = len(roots) // int(cfg["num_roots"])
    data = np.ones(indices.size, dtype=np.float32)
    ps = ref.ParallelSampler(indptr, indices, data, P, threads, True, True, [], 1, "", "", "", seed)
    ps.shuffle_targets(np.asarray(roots, dtype=np.uint32))
    ppr_tab = None
    if ppr_args is not None:
        with tempfile.TemporaryDirectory() as td:
            fn, fs = os.path.join(td, "neighs.bin"), os.path.join(td, "scores.bin")
            ps.preproc_ppr_approximate(np.asarray(ppr_args["targets"], dtype=np.uint32),
                                       int(ppr_args["k"]), float(ppr_args["alpha"]),
                                       float(ppr_args["epsilon"]), fn, fs)
            hdr, nrows = read_ppr_file(fn, False)
            hdr2, srows = read_ppr_file(fs, True)
            assert hdr == hdr2
            ppr_tab = (hdr, nrows, srows)
    scfg = {k: str(v) for k, v in cfg.items()}
    for key in ("add_self_edge", "include_target_conn"):
        if key in cfg:
            scfg[key] = "true" if cfg[key] else "false"
    out = ps.parallel_sampler_ensemble([scfg], [set(aug)])[0]
    n = out.get_num_valid_subg()
    assert n == P
    res = {}
    for f in FIELDS:
        getter = {"node": "node", "edge_index": "edge_index"}.get(f, f)
        vals = getattr(out, f"get_subgraph_{getter}")()[:n]
        dt = np.float32 if f == "ppr" else np.int64
        res[f] = [np.asarray(v, dtype=dt) for v in vals]
    assert ps.get_idx_root() == 0
    return res, ppr_tab


def pack(prefix, res, store):
    for f in FIELDS:
        arrs = res[f]
        off = np.concatenate([[0], np.cumsum([a.size for a in arrs])]).astype(np.int64)
        cat = np.concatenate(arrs) if len(arrs) else np.zeros(0)
        if f == "ppr":
            cat = cat.astype(np.float32)
        else:
            cat = cat.astype(np.int64)
            # everything the reference returns is a uint32 (NodeType); keep it compact
            assert cat.size == 0 or (cat.min() >= 0 and cat.max() <= 0xFFFFFFFF)
            cat = cat.astype(np.uint32)
        store[f"{prefix}_{f}"] = cat
        store[f"{prefix}_{f}_off"] = off


def cases_for(n, max_deg, link=True):
    cs = []
    # deterministic k-hop: full expansion, and a budget that never binds
    for depth in (1, 2):
        for self_e in (False, True):
            cs.append(dict(cfg=dict(method="khop", num_roots=1, depth=depth, budget=-1,
                                    add_self_edge=self_e, include_target_conn=False),
                           aug=["hops"] if depth == 2 else []))
    cs.append(dict(cfg=dict(method="khop", num_roots=1, depth=3, budget=-1, add_self_edge=True,
                            include_target_conn=False), aug=["hops"], max_roots=6))
    cs.append(dict(cfg=dict(method="khop", num_roots=1, depth=2, budget=max_deg + 1,
                            add_self_edge=True, include_target_conn=False), aug=["hops"]))
    cs.append(dict(cfg=dict(method="nodeIID", num_roots=1, add_self_edge=False,
                            include_target_conn=False), aug=[]))
    if link:
        for itc in (False, True):
            cs.append(dict(cfg=dict(method="khop", num_roots=2, depth=1, budget=-1,
                                    add_self_edge=True, include_target_conn=itc), aug=["drnls"]))
        cs.append(dict(cfg=dict(method="khop", num_roots=2, depth=2, budget=-1,
                                add_self_edge=False, include_target_conn=False), aug=["drnls"]))
    return cs


def gen_graph_fixture(name, indptr, indices, seed, n_roots=24, with_ppr=True, link=True):
    rng = np.random.default_rng(seed)
    N = indptr.size - 1
    deg = np.diff(indptr.astype(np.int64))
    store = dict(indptr=indptr, indices=indices)
    cases = []
    ci = 0
    for c in cases_for(N, int(deg.max()), link=link):
        R = c["cfg"]["num_roots"]
        P = min(n_roots, c.get("max_roots", n_roots))
        P -= P % 2
        roots = rng.choice(N, size=P * R, replace=(P * R > N)).astype(np.uint32)
        res, _ = run_reference(indptr, indices, roots, c["cfg"], c["aug"])
        store[f"c{ci}_roots"] = roots
        pack(f"c{ci}", res, store)
        cases.append(dict(idx=ci, **c))
        ci += 1
    if with_ppr:
        targets = rng.choice(N, size=min(N, n_roots), replace=False).astype(np.uint32)
        for (k, thr, eps, self_e) in ((8, 0.0, 1e-4, False), (16, 0.05, 1e-5, True), (64, 0.0, 1e-6, True)):
            cfg = dict(method="ppr", num_roots=1, k=k, threshold=thr, add_self_edge=self_e,
                       include_target_conn=False)
            ppr_args = dict(targets=targets, k=k, alpha=0.85, epsilon=eps)
            res, tab = run_reference(indptr, indices, targets, cfg, ["hops", "pprs"], ppr_args=ppr_args)
            hdr, nrows, srows = tab
            store[f"c{ci}_roots"] = targets
            pack(f"c{ci}", res, store)
            ln = np.array([nrows[t].size for t in targets], dtype=np.uint32)
            nb = np.full((targets.size, k), 0xFFFFFFFF, dtype=np.uint32)
            sc = np.zeros((targets.size, k), dtype=np.float32)
            for i, t in enumerate(targets):
                nb[i, :ln[i]] = nrows[t]; sc[i, :ln[i]] = srows[t]
            # rows of non-targets must be empty (ParallelSampler.cpp:240-241)
            assert all(nrows[v].size == 0 for v in range(N) if v not in set(targets.tolist()))
            store[f"c{ci}_ppr_len"] = ln; store[f"c{ci}_ppr_neigh"] = nb; store[f"c{ci}_ppr_score"] = sc
            store[f"c{ci}_ppr_hdr"] = np.array(hdr, dtype=np.float64)
            cases.append(dict(idx=ci, cfg=cfg, aug=["hops", "pprs"], ppr=dict(k=k, alpha=0.85, epsilon=eps)))
            ci += 1
    store["cases"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", f"sampler_{name}.npz")
    np.savez_compressed(path, **store)
    print(f"wrote {path}: {len(cases)} cases, {os.path.getsize(path)/1024:.1f} KiB")


def gen_budget_stats():
    """Budgeted k-hop in the reference draws from glibc rand(): only the
    distribution can be compared.  Store per-root subgraph sizes over many
    repetitions (1 thread, fixed seeds)."""
    indptr, indices = make_graph(400, 12, seed=11)
    rng = np.random.default_rng(5)
    roots = rng.choice(400, size=40, replace=False).astype(np.uint32)
    cfg = dict(method="khop", num_roots=1, depth=2, budget=4, add_self_edge=False, include_target_conn=False)
    sizes = []
    for rep in range(64):
        res, _ = run_reference(indptr, indices, roots, cfg, [], threads=1, seed=rep)
        sizes.append([a.size for a in res["node"]])
    path = os.path.join(ROOT, "tests", "golden", "sampler_budget_stats.npz")
    np.savez_compressed(path, indptr=indptr, indices=indices, roots=roots,
                        sizes=np.asarray(sizes, dtype=np.int32), depth=2, budget=4)
    print(f"wrote {path}")


def hub_graph_and_roots():
    """The graph and roots of the budget-20 statistics fixture: 5 000 nodes, heavy tail (maximum degree 580 >> budget 20);
    64 roots = the 24 largest hubs (degree 100 .. 580), 24 nodes of degree 21 .. 60 (budget binds, few repeats possible) and
    16 of degree <= 20 (all neighbours taken)."""
    indptr, indices = make_graph(5000, 16, seed=21)
    deg = np.diff(indptr.astype(np.int64))
    rng = np.random.default_rng(23)
    hubs = np.argsort(-deg, kind="stable")[:24]
    mid = rng.choice(np.nonzero((deg > 20) & (deg <= 60))[0], size=24, replace=False)
    low = rng.choice(np.nonzero((deg >= 3) & (deg <= 20))[0], size=16, replace=False)
    roots = np.concatenate([hubs, mid, low]).astype(np.uint32)
    return indptr, indices, roots


def gen_budget_hub_stats(reps=256):
    """(round 5) The draw statistics of the reference's budgeted k-hop where the budget BINDS hard --
    budget 20 on roots of degree up to 580, as in every k-hop BASELINE config (products: mean degree 50) -- over `reps`
    one-thread runs with seeds 0 .. reps-1: per (root, node) how many runs' subgraphs contain the node, and every run's
    subgraph sizes, at depth 1 (exact theory available: 20 uniform draws WITH replacement from the root's row,
    ParallelSampler.cpp:528-540) and depth 2.  Data only."""
    indptr, indices, roots = hub_graph_and_roots()
    N = indptr.size - 1
    store = dict(indptr=indptr, indices=indices, roots=roots, budget=20, reps=reps)
    for depth in (1, 2):
        cfg = dict(method="khop", num_roots=1, depth=depth, budget=20, add_self_edge=False, include_target_conn=False)
        counts = np.zeros((roots.size, N), dtype=np.uint16)
        sizes = np.zeros((reps, roots.size), dtype=np.uint16)
        for rep in range(reps):
            res, _ = run_reference(indptr, indices, roots, cfg, [], threads=1, seed=rep)
            for r, nodes in enumerate(res["node"]):
                counts[r, nodes] += 1
                sizes[rep, r] = nodes.size
        store[f"d{depth}_counts"] = counts
        store[f"d{depth}_sizes"] = sizes
    path = os.path.join(ROOT, "tests", "golden", "sampler_budget_hub_stats.npz")
    np.savez_compressed(path, **store)
    print(f"wrote {path}: {os.path.getsize(path)/1024:.1f} KiB")


def gen_ppr_cache_files():
    """The reference's two PPR cache files as RAW BYTES (ParallelSampler.cpp:94-137, written by
    preproc_ppr_approximate :344) for a small graph: the byte-compatibility fixture of sg_load_ppr_bin /
    sg_save_ppr_bin.  Data only: graph, targets, parameters, file contents."""
    import ParallelSampler as ref  # oracle/_ref
    indptr, indices = make_graph(120, 6, seed=13)
    rng = np.random.default_rng(17)
    targets = np.sort(rng.choice(120, size=20, replace=False)).astype(np.uint32)
    k, alpha, eps = 12, 0.85, 1e-4
    ps = ref.ParallelSampler(indptr, indices, np.ones(indices.size, dtype=np.float32), 20, 1, True, True, [], 1, "", "", "", 0)
    with tempfile.TemporaryDirectory() as td:
        fn, fs = os.path.join(td, "neighs.bin"), os.path.join(td, "scores.bin")
        ps.preproc_ppr_approximate(targets, k, alpha, eps, fn, fs)
        bn, bs = open(fn, "rb").read(), open(fs, "rb").read()
        # a second sampler object accepts its own files (the reader's acceptance rule, .cpp:166) ...
        ps2 = ref.ParallelSampler(indptr, indices, np.ones(indices.size, dtype=np.float32), 20, 1, True, True, [], 1, "", "", "", 0)
        ps2.preproc_ppr_approximate(targets, 8, alpha, eps * 1.05, fn, fs)         # smaller k, eps within 10 %: loaded, clipped
        ps2.shuffle_targets(targets)
        cfg = {"method": "ppr", "k": "8", "num_roots": "1", "threshold": "0.0", "add_self_edge": "true",
               "include_target_conn": "false"}
        out = ps2.parallel_sampler_ensemble([cfg], [{"pprs"}])[0]
        nodes = out.get_subgraph_node()[:out.get_num_valid_subg()]
        pprs = out.get_subgraph_ppr()[:out.get_num_valid_subg()]
        assert open(fn, "rb").read() == bn                                          # a successful load does not rewrite
    off = np.concatenate([[0], np.cumsum([len(v) for v in nodes])]).astype(np.int64)
    path = os.path.join(ROOT, "tests", "golden", "ppr_cache_files.npz")
    np.savez_compressed(path, indptr=indptr, indices=indices, targets=targets, k=k, alpha=alpha, epsilon=eps,
                        neighs_bytes=np.frombuffer(bn, dtype=np.uint8), scores_bytes=np.frombuffer(bs, dtype=np.uint8),
                        clip_k=8, clip_node=np.concatenate([np.asarray(v, dtype=np.uint32) for v in nodes]),
                        clip_ppr=np.concatenate([np.asarray(v, dtype=np.float32) for v in pprs]), clip_off=off)
    print(f"wrote {path}: {len(bn)} + {len(bs)} file bytes")


def main():
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    if "--ppr-files-only" in sys.argv:
        gen_ppr_cache_files()
        return
    ip, ix = path_graph()
    gen_graph_fixture("path6", ip, ix, seed=1, n_roots=6, with_ppr=True, link=True)
    ip, ix = make_graph(300, 8, seed=3)
    gen_graph_fixture("rand300", ip, ix, seed=2)
    ip, ix = make_graph(500, 10, seed=7, self_loops=25)
    gen_graph_fixture("selfloop500", ip, ix, seed=4)
    ip, ix = make_graph(400, 6, seed=9, directed=True)
    gen_graph_fixture("directed400", ip, ix, seed=6, link=False)
    gen_budget_stats()
    gen_budget_hub_stats()
    gen_ppr_cache_files()


if __name__ == "__main__":
    main()
