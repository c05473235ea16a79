"""Fuzz of DeepGNN (HIP path) against oracle/layers_oracle.model_forward: random architecture (family, depth,
width, heads, activation, residue, pooling, hop augmentation), block-diagonal batches; predictions, loss and
every parameter gradient.  python scripts/fuzz_models.py [seed] [trials]"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
DEV = "cuda:0"


def run(seed: int, trials: int, verbose: bool = True):
    from oracle import layers_oracle as lo
    from shadow_gnn_amd import ops
    from shadow_gnn_amd.minibatch import TRAIN, OneBatchSubgraph
    from shadow_gnn_amd.models import DeepGNN
    rng = np.random.default_rng(seed)
    failures = []
    for trial in range(trials):
        aggr = str(rng.choice(["sage", "gcn", "gat"]))
        heads = int(rng.choice([1, 2, 4])) if aggr == "gat" else 1
        dim = heads * int(rng.choice([4, 8, 16, 25])) if aggr == "gat" else int(rng.choice([8, 16, 47, 64]))
        arch = dict(num_layers=int(rng.integers(1, 5)), num_cls_layers=1, heads=heads, branch_sharing=False, dim=dim,
                    act=str(rng.choice(["relu", "elu", "tanh", "prelu"])), layer_norm="norm_feat", feature_augment_ops="sum",
                    aggr=aggr, residue=str(rng.choice(["none", "none", "max", "sum", "concat"])),
                    pooling=str(rng.choice(["center", "center", "mean", "max", "sum"])), loss="softmax", ensemble_act="relu")
        aug = bool(rng.random() < 0.4)
        B = int(rng.choice([1, 3, 9])); F0 = int(rng.choice([5, 12, 33])); C = int(rng.choice([2, 5, 11]))
        sizes = rng.integers(1, 40, B)
        blocks = []
        for s_ in sizes:
            a = (rng.random((s_, s_)) < rng.choice([0.05, 0.3])).astype(np.float32)
            a = np.maximum(a, a.T)
            if aggr == "gcn" or rng.random() < 0.5:
                np.fill_diagonal(a, 1.0)
            blocks.append(sp.csr_matrix(a))
        A = sp.block_diag(blocks, format="csr"); A.sort_indices()
        n = A.shape[0]
        off = np.concatenate([[0], np.cumsum(sizes)])
        target = (off[:-1] + rng.integers(0, sizes)).astype(np.int64)
        ctx = (trial, {k: arch[k] for k in ("aggr", "num_layers", "dim", "heads", "act", "residue", "pooling")}, aug, B, n)
        try:
            torch.manual_seed(trial)
            aug_feat = [("hops", 7)] if aug else []
            m = DeepGNN(F0, F0, C, 0, arch, aug_feat, 1, dict(lr=0.01, dropout=0.0, dropedge=0.0), "node").to(DEV)
            with torch.no_grad():
                for q in m.parameters():
                    q.add_(0.1 * torch.randn_like(q))
            X = torch.randn(n, F0)
            labels = rng.integers(0, C, B)
            hop = rng.integers(0, 7, n)
            hop1hot = torch.nn.functional.one_hot(torch.as_tensor(hop), 7).float() if aug else None
            csr = ops.DeviceCSR(torch.from_numpy(A.indptr.astype(np.int32)).to(DEV), torch.from_numpy(A.indices.astype(np.int32)).to(DEV))
            bt = OneBatchSubgraph([csr], [X.to(DEV)], torch.as_tensor(labels).to(DEV), torch.as_tensor(sizes.astype(np.int64)).to(DEV).unsqueeze(0),
                                  [torch.as_tensor(target).to(DEV)], [{"hops": hop1hot.to(DEV)} if aug else {}])
            m.train()
            preds, _ = m(TRAIN, dropedge=0.0, **bt.to_dict({"feat_ens", "adj_ens", "target_ens", "size_subg_ens", "feat_aug_ens"}))
            loss = m._loss(preds, torch.nn.functional.one_hot(torch.as_tensor(labels).to(DEV), C))
            loss.backward()
            p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
            ref, _ = lo.model_forward(p, arch, X, A.indptr, A.indices, sizes, target, hop1hot)
            rl = lo.model_loss(ref, labels)
            rl.backward()
            np.testing.assert_allclose(preds.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-3, atol=1e-3)
            assert abs(float(loss.detach()) - float(rl.detach())) < 1e-3
            for k, q in m.named_parameters():
                if p[k].grad is None:
                    assert q.grad is None or float(q.grad.abs().max()) == 0.0, k
                    continue
                np.testing.assert_allclose(q.grad.cpu().numpy(), p[k].grad.numpy(), rtol=2e-2, atol=2e-3, err_msg=k)
            if m._tail_prunable(0):            # the target-only tail must reproduce the full stack on the same batch
                full = {k: q.grad.clone() for k, q in m.named_parameters() if q.grad is not None}
                m.zero_grad(set_to_none=True)
                m.prune_tail = True
                bt2 = OneBatchSubgraph([csr], [X.to(DEV)], bt.label, bt.size_subg_ens, bt.target_ens, bt.feat_aug_ens)
                preds2, _ = m(TRAIN, dropedge=0.0, **bt2.to_dict({"feat_ens", "adj_ens", "target_ens", "size_subg_ens", "feat_aug_ens"}))
                m._loss(preds2, torch.nn.functional.one_hot(torch.as_tensor(labels).to(DEV), C)).backward()
                np.testing.assert_allclose(preds2.detach().cpu().numpy(), preds.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
                for k, q in m.named_parameters():
                    if k in full:
                        np.testing.assert_allclose(q.grad.cpu().numpy(), full[k].cpu().numpy(), rtol=1e-3, atol=1e-4, err_msg="tail " + k)
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
