"""Sampler kernel time vs roots per call (products shape, k-hop depth 2 budget 20): how much of the per-call
time is per-subgraph latency (1024 subgraphs = exactly two rounds of 512 resident workgroups) and how much is
throughput."""
import sys, numpy as np, torch
from shadow_gnn_amd.sampler import HipSampler, SamplerConfig
from shadow_gnn_amd.synthetic import SHAPES, MAX_DEGREE, make_graph_torch
dev = torch.device("cuda:0")
import os
SHAPE = os.environ.get("SHAPE", "products")
N, nnz, F, C = SHAPES[SHAPE]
indptr, indices = make_graph_torch(N, nnz, seed=0, device=dev, max_degree=MAX_DEGREE[SHAPE])
hs = HipSampler(indptr, indices, device=dev, seed=3)
roots = torch.randperm(N, generator=torch.Generator().manual_seed(2)).numpy().astype(np.uint32)
hs.shuffle_targets(roots)
hs.set_profiling(True)
cfg = SamplerConfig(method="khop", depth=int(os.environ.get("DEPTH", "2")), budget=int(os.environ.get("BUDGET", "20")),
                    add_self_edge=os.environ.get("SELF", "0") == "1")
Bs = [int(a) for a in sys.argv[1:]] or [256, 512, 768, 1024, 1536, 2048, 4096, 8192]
for B in Bs:
    ms, nn, slots = [], 0, 0
    for it in range(6):
        b = hs.sample(cfg, B)
        if it >= 2:
            ms.append(b.counts["sample_kernel_ms"]); nn += b.num_nodes; slots += b.counts["slots_scanned"]
    m = float(np.mean(ms))
    import ctypes as C
    ph = (C.c_uint32 * 24)()
    hs._lib.sg_debug_scan_phases(hs._h, ph)
    tot = sum(ph[8:13]) or 1
    print(f"   scan: chunks {ph[0]} per workgroup {ph[1]} segments {ph[13]} rounds {ph[14]}  phases setup/startrows/scan/resolve/sortwrite = "
          + " / ".join(f"{100 * ph[8 + i] / tot:.0f}%" for i in range(5)) + f"   avg cycles per segment {16 * tot / max(1, ph[13]):.0f}"
          + f"\n         wave 0 (cycles per segment): run streaming {16 * ph[15] / max(1, ph[13]):.0f}, short rows {16 * ph[16] / max(1, ph[13]):.0f}, "
          f"barrier wait {16 * ph[10] / max(1, ph[13]):.0f}; bucket scan {16 * ph[17] / max(1, ph[13]):.0f}, rank {16 * ph[18] / max(1, ph[13]):.0f}, "
          f"ticket wait {16 * ph[19] / max(1, ph[13]):.0f}, write-out {16 * ph[12] / max(1, ph[13]):.0f}")
    print(f"   scan workgroups: longest {16 * ph[16]:.0f} cycles, mean {16 * tot / 256:.0f} (256 workgroups assumed: one per CU for node sets beyond the LDS tables)")
    print(f"   select (cycles per subgraph): expansion {16 * ph[20] / B:.0f}, sort {16 * ph[21] / B:.0f}, row records {16 * ph[22] / B:.0f}")
    print(f"B={B:5d}: sample kernel {m:.3f} ms  {nn / 4 / m / 1e3:.0f} M nodes/s  neighbour ids scanned {slots / 4 * 4 / m / 1e6:.0f} GB/s  "
          f"({m / B * 1e3:.3f} us per subgraph)")
