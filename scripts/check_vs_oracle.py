"""Compare the HIP sampler with the oracle on a full-size synthetic graph (GPU box)."""
import sys, time, numpy as np, torch
from oracle import sampler_oracle as so
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
shape = sys.argv[1] if len(sys.argv) > 1 else "products"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
selfe = int(sys.argv[3]) if len(sys.argv) > 3 else 0
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda:0")
N, nnz, F, C = SHAPES[shape]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE[shape])
ip, ix = indptr.cpu().numpy(), indices.cpu().numpy()
so.build()
hs = HipSampler(indptr, indices, device=dev, seed=3)
roots = torch.randperm(N, generator=torch.Generator().manual_seed(2))[:B].numpy().astype(np.uint32)
hs.shuffle_targets(roots)
cfg = SamplerConfig(method="khop", depth=depth, budget=20, add_self_edge=bool(selfe), aug=("hops",))
b = hs.sample(cfg, B)
h = b.to_host()
t0 = time.time()
ref = so.sample_batch(ip, ix, roots, method="khop", depth=depth, budget=20, add_self_edge=bool(selfe), aug=("hops",),
                      seed=3, serial_base=0, num_threads=32)
print("oracle %.1fs  n=%d e=%d | hip n=%d e=%d" % (time.time() - t0, len(ref.node), len(ref.indices), b.num_nodes, b.num_edges))
ok = True
for k in ("node", "indptr", "indices", "edge_id", "target", "hop"):
    same = np.array_equal(h[k], getattr(ref, k))
    ok &= same
    print(k, "OK" if same else "DIFF")
if not ok:
    hn = h["subg_node_off"].astype(np.int64); he = h["subg_edge_off"].astype(np.int64)
    rn = np.concatenate([[0], np.cumsum(ref.subg_nodes, dtype=np.int64)])
    re_ = np.concatenate([[0], np.cumsum(ref.subg_edges, dtype=np.int64)])
    dn = np.diff(hn) - np.diff(rn); de = np.diff(he) - np.diff(re_)
    bad = np.nonzero((dn != 0) | (de != 0))[0]
    print("subgraphs with different sizes:", bad[:20], "dn", dn[bad[:20]], "de", de[bad[:20]])
    for sidx in bad[:3]:
        n0, n1 = rn[sidx], rn[sidx + 1]
        rip = ref.indptr[n0:n1 + 1] ; hip_ = h["indptr"][hn[sidx]:hn[sidx + 1] + 1]
        rdeg = np.diff(rip.astype(np.int64)); hdeg = np.diff(hip_.astype(np.int64))
        rows = np.nonzero(rdeg != hdeg)[0]
        nodes = ref.node[n0:n1]
        print(" subgraph", sidx, "n", n1 - n0, "rows differing", rows[:10], "ref deg", rdeg[rows[:10]], "hip deg", hdeg[rows[:10]],
              "full deg", (ip[nodes[rows[:10]] + 1] - ip[nodes[rows[:10]]]), "row start mod 4", ip[nodes[rows[:10]]] % 4)
        e0 = ip[nodes].astype(np.int64); e1 = ip[nodes + 1].astype(np.int64)
        vq = np.where(e1 > e0, ((e1 - 1) >> 2) - (e0 >> 2) + 1, 0)
        qp = np.concatenate([[0], np.cumsum(vq)])
        print("   Q", qp[-1], "qptr of differing rows", qp[rows[:10]], "..", qp[rows[:10] + 1], " zero-degree rows:", np.nonzero(vq == 0)[0][:10])
sys.exit(0 if ok else 1)
