"""Fuzz of the pooled read-out's gradient table (ops.POOL_GRAD_TABLE) on whole train steps: random depth, width, activation, residue,
pooling (mean / sum), dropout, drop-edge, batch size, frozen layers -- the step with the table must be BIT-IDENTICAL to the step with the
dense [n, F] hand-over (loss, predictions, every parameter gradient), and within 5e-6 of the un-chained round-5 path (development aid).
    python scripts/fuzz_pool_table.py [seed] [trials]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from shadow_gnn_amd import ops
from tests.test_layers_gpu import _sage_stack_step

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(seed)
bad = 0
for t in range(trials):
    cfg = dict(n_layers=int(rng.integers(1, 6)), dim=int(rng.choice([160, 192, 224, 256, 256])), p_drop=float(rng.choice([0.0, 0.1, 0.3, 0.5])),
               act=str(rng.choice(["relu", "elu", "tanh"])), residue=str(rng.choice(["none", "max", "concat", "sum"])),
               pooling=str(rng.choice(["mean", "sum", "mean", "max"])), dropedge=float(rng.choice([0.0, 0.1])), B=int(rng.choice([24, 64, 96])),
               F0=int(rng.choice([100, 128])))
    freeze = (0,) if rng.random() < 0.15 and cfg["n_layers"] > 1 else ()
    res, calls = {}, {}
    try:
        for name, (tab, chain_dual) in dict(table=(True, True), dense=(False, True), unchained=(False, False)).items():
            pt, pc = ops.POOL_GRAD_TABLE, ops.CHAIN_DUAL
            ops.POOL_GRAD_TABLE, ops.CHAIN_DUAL = tab, chain_dual
            c0 = ops._PoolAndRoots.table_calls
            try:
                res[name] = _sage_stack_step(cfg["n_layers"], cfg["dim"], cfg["p_drop"], 100 + t, chain=True, fused=True, B=cfg["B"], act=cfg["act"],
                                             F0=cfg["F0"], dropedge=cfg["dropedge"], pooling=cfg["pooling"], residue=cfg["residue"], freeze=freeze)
            finally:
                ops.POOL_GRAD_TABLE, ops.CHAIN_DUAL = pt, pc
            calls[name] = ops._PoolAndRoots.table_calls - c0
        a, b, c = res["table"], res["dense"], res["unchained"]
        assert a[0] == b[0] and torch.equal(a[1], b[1]), "loss / predictions differ"
        for k in a[2]:
            assert torch.equal(a[2][k], b[2][k]), f"gradient {k} differs (max {float((a[2][k] - b[2][k]).abs().max()):.3e})"
        for k in a[2]:
            sc = float(c[2][k].abs().max())
            assert float((a[2][k] - c[2][k]).abs().max()) <= 5e-6 * sc + 1e-9, f"gradient {k} against the un-chained path"
        print("ok ", t, cfg, "frozen" if freeze else "", "table nodes", calls["table"]); sys.stdout.flush()
    except Exception as ex:
        bad += 1
        print("BAD", t, cfg, freeze, type(ex).__name__, str(ex)[:300].replace("\n", " ")); sys.stdout.flush()
print("done", trials, "trials,", bad, "bad")
