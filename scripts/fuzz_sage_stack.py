"""Fuzz of the whole-stack nodes (GraphSAGE: ops._SageStack, sl_sage_stack_fwd / sl_sage_stack_bwd; GCN: ops._GcnStack, sl_gcn_stack_*) against the layer-by-layer nodes:
random depth, width, input width, activation, dropout, drop-edge, augmentation, batch size -- loss, predictions and every
parameter gradient must be bit-identical (the C entries run the same per-layer entries in the same order), in training and in
evaluation mode.  python scripts/fuzz_sage_stack.py [seed] [trials]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(seed, trials):
    import test_layers_gpu as T
    from shadow_gnn_amd import ops
    rng = np.random.default_rng(seed)
    bad = 0
    for t in range(trials):
        kw = dict(n_layers=int(rng.integers(1, 6)), dim=int(rng.choice([32, 64, 128, 256])), p_drop=float(rng.choice([0.0, 0.2, 0.5])),
                  seed=int(rng.integers(1, 1000)), chain=True, fused=True, B=int(rng.choice([16, 64, 128, 300])),
                  act=str(rng.choice(["relu", "elu", "tanh", "leakyrelu"])), F0=int(rng.choice([36, 100, 128, 256])), sparse_top=False,
                  dropedge=float(rng.choice([0.0, 0.1])), aug=bool(rng.random() < 0.4), aggr=str(rng.choice(["sage", "sage", "gcn"])))
        ok = True
        calls = lambda: ops._SageStack.calls + ops._GcnStack.calls
        for train in (True, False):
            k0 = calls()
            a = T._sage_stack_step(stack=False, train=train, **kw)
            k1 = calls()
            b = T._sage_stack_step(stack=True, train=train, **kw)
            took = calls() - k1
            same = a[0] == b[0] and torch.equal(a[1], b[1]) and set(a[2]) == set(b[2]) and all(torch.equal(a[2][k], b[2][k]) for k in a[2])
            ok = ok and same and k1 == k0
            if not same or k1 != k0:
                print("MISMATCH", kw, "train" if train else "eval", "stack taken", took, flush=True)
            elif took != 1:
                print("  (stack not taken:", kw, ")", flush=True)
        bad += 0 if ok else 1
    print("done", trials, "trials,", bad, "bad")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 30) else 0)
