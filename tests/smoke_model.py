"""__graft_entry__.smoke(): one tiny sample -> gather -> SAGE-3 train step on cuda:0, forward
checked against the CPU oracle (tolerance 1e-4).  Test infrastructure (it imports oracle/): lives
under tests/, not in the product package."""
import numpy as np
import torch


def run(hs, cfg):
    from oracle import layers_oracle as lo
    from shadow_gnn_amd.minibatch import OneBatchSubgraph, TRAIN, hop2onehot
    from shadow_gnn_amd.models import DeepGNN
    from shadow_gnn_amd import ops
    dev = hs.device
    torch.manual_seed(0)
    N, F0, C = hs.num_nodes(), 16, 5
    feat_full = torch.randn(N, F0, device=dev)
    roots = np.random.default_rng(7).permutation(N)[:32].astype(np.uint32)
    b = hs.sample(cfg, roots=roots, serial_base=100)
    arch = dict(num_layers=3, num_cls_layers=1, heads=1, dim=32, act="elu", layer_norm="norm_feat",
                feature_augment_ops="sum", aggr="sage", residue="none", pooling="center", loss="softmax")
    model = DeepGNN(F0, F0, C, 0, arch, [("hops", 7)], 1, dict(dropout=0.0, dropedge=0.0, lr=1e-3), "node").to(dev)
    model.optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    labels = torch.randint(0, C, (32,), device=dev)
    hop1 = hop2onehot(b.hop, 7)
    X = ops.gather_rows(feat_full, b.node)
    h = b.to_host()
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref_preds, _ = lo.model_forward(params, arch, X.cpu(), h["indptr"], h["indices"],
                                    np.diff(h["subg_node_off"].astype(np.int64)), h["target"], hop1.cpu())
    batch = OneBatchSubgraph([ops.DeviceCSR(b.indptr, b.indices)], [X], labels, b.size_subg.unsqueeze(0),
                             [b.target], [{"hops": hop1}])
    ret = model.step(TRAIN, "running", batch)
    got = ret["preds"].detach().cpu()
    ref = torch.softmax(ref_preds, dim=1)
    err = float((got - ref).abs().max())
    assert err < 1e-4, f"smoke: model forward differs from the oracle by {err}"
    print(f"[smoke] SAGE-3 train step ok: loss={float(ret['loss']):.4f} max|preds-oracle|={err:.2e}")
