"""Quick sampler-only timing probe (development aid, not the bench contract)."""
import argparse, time
import numpy as np, torch
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from shadow_gnn_amd.synthetic import SHAPES, make_graph_torch

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="products")
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--depth", type=int, default=2)
ap.add_argument("--budget", type=int, default=20)
ap.add_argument("--self-edge", type=int, default=0)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--uncapped", type=int, default=0)
a = ap.parse_args()
UNCAP, SHAPE = a.uncapped, a.shape
dev = torch.device("cuda:0")
N, nnz, F, Cc = SHAPES[a.shape]
t0 = time.time()
from shadow_gnn_amd.synthetic import MAX_DEGREE
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=(None if UNCAP else MAX_DEGREE[SHAPE]))
torch.cuda.synchronize()
print(f"graph {a.shape}: N={N} nnz={indices.numel()} gen {time.time()-t0:.1f}s")
hs = HipSampler(indptr, indices, device=dev, seed=3)
roots = torch.randperm(N, generator=torch.Generator().manual_seed(2))[: a.batch * (a.iters + 5)].numpy().astype(np.uint32)
hs.shuffle_targets(roots)
cfg = SamplerConfig(method="khop", depth=a.depth, budget=a.budget, add_self_edge=bool(a.self_edge))
for _ in range(5):
    b = hs.sample(cfg, a.batch)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot_n = tot_e = tot_slots = tot_fr = 0
t0 = time.time(); ev0.record()
for _ in range(a.iters):
    b = hs.sample(cfg, a.batch)
    tot_n += b.num_nodes; tot_e += b.num_edges; tot_slots += b.counts["slots_scanned"]; tot_fr += b.counts["frontier_reads"]
ev1.record(); torch.cuda.synchronize()
dt = time.time() - t0
ms = ev0.elapsed_time(ev1) / a.iters
n, e, sl = tot_n / a.iters, tot_e / a.iters, tot_slots / a.iters
byt = 4 * sl + 8 * n + 4 * n + 4 * (n + 1) + 8 * e + 4 * tot_fr / a.iters
print(f"per call: {ms:.3f} ms (wall {dt/a.iters*1e3:.3f} ms)  n={n:.0f} e={e:.0f} slots={sl:.0f}  "
      f"nodes/s={n/(dt/a.iters):.3e}  alg GB/s={byt/(ms*1e-3)/1e9:.1f}  max_n={b.counts['max_subg_nodes']} max_e={b.counts['max_subg_edges']}")

# ---- steady state: two sampler handles alternate so the GPU always has the next call queued
# (a lone synchronous call per iteration leaves idle gaps and the clocks sag)
hs2 = HipSampler(indptr, indices, device=dev, seed=3)
hs2.shuffle_targets(roots)
for h in (hs, hs2):
    h.set_profiling(True)
hs.sample_async(cfg, a.batch); hs2.sample_async(cfg, a.batch)
ks, kr = [], []
torch.cuda.synchronize(); t0 = time.time()
for it in range(a.iters):
    for h in (hs, hs2):
        b = h.finish()
        ks.append(b.counts["sample_kernel_ms"]); kr.append(b.counts["relocate_kernel_ms"])
        h.sample_async(cfg, a.batch)
torch.cuda.synchronize(); dt = time.time() - t0
hs.finish(); hs2.finish()
ks, kr = np.array(ks[4:]), np.array(kr[4:])
print(f"pipelined: {dt/(2*a.iters)*1e3:.3f} ms/call wall; sample kernel {ks.mean():.3f} ms (min {ks.min():.3f}), relocate {kr.mean():.3f} ms "
      f"-> sample-kernel alg GB/s {byt/(ks.mean()*1e-3)/1e9:.1f}")
