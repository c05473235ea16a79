"""Fuzz of the HIP layers against the dense layer oracle: random widths (odd, tiny, > 256), activations, head
counts, graphs with isolated rows, hubs, directed edges; outputs and all gradients.
    python scripts/fuzz_layers.py [seed] [trials]
(tests/test_layers_gpu.py runs one seed as a regression test.)  A failure is not necessarily a bug: an activation
kink hit within rounding (|z| ~ 1e-8 under relu / prelu) flips one derivative and shows up in a whole row of dX."""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
DEV = "cuda:0"


def run(seed: int, trials: int, verbose: bool = True):
    from oracle import layers_oracle as lo
    from shadow_gnn_amd import layers, ops
    rng = np.random.default_rng(seed)
    failures = []
    for trial in range(trials):
        kind = str(rng.choice(["sage", "gcn", "gat"]))
        n = int(rng.choice([1, 2, 7, 60, 300, 900]))
        F_in = int(rng.choice([1, 3, 4, 17, 64, 100, 257]))
        act = str(rng.choice(["relu", "elu", "tanh", "leakyrelu", "I", "prelu", "prelu+"]))
        heads = int(rng.choice([1, 2, 3, 4])) if kind == "gat" else 1
        # (normalising 1-2 features is ill-conditioned: var + 1e-9)
        F_out = heads * int(rng.choice([4, 5, 16, 64, 100])) if kind == "gat" else int(rng.choice([5, 16, 47, 128, 256, 300]))
        dens = float(rng.choice([0.0, 0.01, 0.2]))
        a = (rng.random((n, n)) < dens).astype(np.float32)
        if rng.random() < 0.5:
            a = np.maximum(a, a.T)
        if rng.random() < 0.5:
            np.fill_diagonal(a, 1.0)
        if n > 5 and rng.random() < 0.4:
            a[0, :] = 1.0                      # a hub row
        A = sp.csr_matrix(a); A.sort_indices()
        ctx = (trial, kind, n, F_in, F_out, heads, act, dens)
        try:
            torch.manual_seed(trial)
            cls = {"sage": layers.GraphSAGE, "gcn": layers.GCN, "gat": layers.GAT}[kind]
            layer = cls(F_in, F_out, dropout=0.0, act=act, norm="norm_feat", mulhead=heads).to(DEV)
            with torch.no_grad():
                for q in layer.parameters():
                    q.add_(0.1 * torch.randn_like(q))
            X = torch.randn(n, F_in); G = torch.randn(n, F_out)
            x = X.to(DEV).requires_grad_(True)
            csr = ops.DeviceCSR(torch.from_numpy(A.indptr.astype(np.int32)).to(DEV), torch.from_numpy(A.indices.astype(np.int32)).to(DEV))
            out, _, _, _ = layer((x, csr, False, 0.0), None)
            (out * G.to(DEV)).sum().backward()
            p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
            xr = X.clone().requires_grad_(True)
            ref = lo.layer_forward(kind, p, xr, A.indptr, A.indices, act, heads=heads)
            (ref * G).sum().backward()
            np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-4, atol=2e-4)
            np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-3, atol=5e-4)
            for k, q in layer.named_parameters():
                np.testing.assert_allclose(q.grad.cpu().numpy(), p[k].grad.numpy(), rtol=5e-3, atol=2e-3, err_msg=k)
        except Exception as ex:
            failures.append((ctx, type(ex).__name__, str(ex)[:300].replace("\n", " ")))
            if verbose:
                print("BAD", ctx, type(ex).__name__, str(ex)[:300].replace("\n", " ")); sys.stdout.flush()
    return failures


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    f = run(seed, trials)
    print("done", trials, "trials,", len(f), "bad")
