"""papers100M-shape (N = 111 M nodes, ~3.2 G directed edges) sampler run on ONE MI355X: the whole CSR
(13.4 GB) lives in HBM.  Times k-hop and PPR sampling and checks size-independent properties of the output
(development aid / evidence for BASELINE config 5; not part of the bench contract)."""
import sys, time, numpy as np, torch
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
from shadow_gnn_amd.ppr import ppr_approximate_device
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda:0")
N, nnz, F, C = SHAPES["papers100M"]
nnz = int(nnz * scale)
t0 = time.time()
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE["papers100M"])
torch.cuda.synchronize()
print(f"graph: N={N} nnz={indices.numel()} ({(indptr.numel()+indices.numel())*4/1e9:.1f} GB CSR) generated in {time.time()-t0:.1f}s, "
      f"peak HBM {torch.cuda.max_memory_allocated()/1e9:.1f} GB", flush=True)
torch.cuda.empty_cache()
hs = HipSampler(indptr, indices, device=dev, seed=3)
B = 2048
roots = torch.randperm(N, generator=torch.Generator().manual_seed(2))[:B * 12].numpy().astype(np.uint32)
hs.shuffle_targets(roots)


def check(b, tag):
    """size-independent properties: per subgraph sorted unique node ids; every edge's edge_id addresses the
    full-graph entry (row = source node, column = destination node); targets map back to the roots."""
    node = b.node.long(); ip = b.indptr.long() & 0xFFFFFFFF; ix = b.indices.long(); eid = b.edge_id.long() & 0xFFFFFFFF
    off = b.subg_node_off.long()
    d = node[1:] - node[:-1]
    starts = torch.zeros(node.numel(), dtype=torch.bool, device=dev); starts[off[1:-1]] = True
    assert bool(((d > 0) | starts[1:]).all()), "node ids not sorted/unique inside a subgraph"
    rows = torch.repeat_interleave(torch.arange(node.numel(), device=dev), ip[1:] - ip[:-1])
    real = eid != 0xFFFFFFFF
    gi = indices.long() & 0xFFFFFFFF
    assert bool((gi[eid[real]] == node[ix[real]]).all()), "edge_id does not address the destination column"
    gp = indptr.long() & 0xFFFFFFFF
    src = node[rows[real]]
    assert bool(((eid[real] >= gp[src]) & (eid[real] < gp[src + 1])).all()), "edge_id outside the source row"
    print(f"  [{tag}] properties ok: n={b.num_nodes} e={b.num_edges} max_n={b.counts['max_subg_nodes']}", flush=True)


for name, cfg in (("khop d2 b20", SamplerConfig(method="khop", depth=2, budget=20)),):
    b = hs.sample(cfg, B); check(b, name)
    torch.cuda.synchronize(); t0 = time.time(); n = 0
    for _ in range(5):
        b = hs.sample(cfg, B); n += b.num_nodes
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f"{name}: {dt*1e3:.2f} ms per call of {B} roots, {n/5/dt/1e6:.1f} M sampled nodes/s, slots/call {b.counts['slots_scanned']/1e6:.1f} M", flush=True)
# PPR: table for the next roots, then top-k=200 sampling
uniq = np.unique(roots[: B * 2])
t0 = time.time()
ln, nb, sc = ppr_approximate_device(hs, uniq, 200, 0.85, 1e-5)
torch.cuda.synchronize(); tp = time.time() - t0
print(f"ppr push: {uniq.size} targets in {tp:.2f}s = {uniq.size/tp:.0f} targets/s", flush=True)
hs.set_ppr(uniq, ln, nb, sc)
hs.shuffle_targets(roots[: B * 2])
cfg = SamplerConfig(method="ppr", k=200, threshold=0.0, add_self_edge=False)
print("ppr caps", hs.get_caps(cfg), flush=True)
b = hs.sample(cfg, B); check(b, "ppr k200")
torch.cuda.synchronize(); t0 = time.time()
hs.shuffle_targets(roots[: B * 2])
b = hs.sample(cfg, B); b2 = hs.sample(cfg, B)
torch.cuda.synchronize(); dt = (time.time() - t0) / 2
print(f"ppr k200: {dt*1e3:.2f} ms per call of {B} roots, {(b.num_nodes+b2.num_nodes)/2/dt/1e6:.1f} M sampled nodes/s", flush=True)
