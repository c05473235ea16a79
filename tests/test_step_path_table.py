"""ops.step_path -- the (layer kind, rows, width, layers, training, read-out) -> path table DeepGNN dispatches on -- as a pure
function (no GPU): the thresholds it reads, the labels it returns.  tests/test_layers_gpu.py::test_step_path_table holds the
table against what actually runs."""
import pytest

from shadow_gnn_amd import ops


def test_small_batches_run_kernel_by_kernel():
    for kind in ("sage", "gcn", "gat"):
        assert ops.step_path(kind, ops.GEMM_SPLIT_MIN_ROWS - 1, 256, 3, True, "center", heads=4) == ops.StepPath("kernels", "kernels")
        assert ops.step_path(kind, 10, 256, 3, False, "mean", heads=4) == ops.StepPath("kernels", "none")


def test_graphsage_rows(monkeypatch):
    big = ops.SPARSE_TOP_BWD_MIN_ROWS
    assert ops.step_path("sage", big, 256, 5, True, "center") == ops.StepPath("stack", "stack+sparse-top")          # the headline step
    assert ops.step_path("sage", big - 1, 256, 5, True, "center") == ops.StepPath("stack", "stack")
    assert ops.step_path("sage", big, 256, 2, True, "center") == ops.StepPath("layer-calls", "chained+sparse-top")  # too few layers for the node's pass
    assert ops.step_path("sage", big, 64, 5, True, "center") == ops.StepPath("layer-calls", "chained+sparse-top")   # under the block-diagonal width
    assert ops.step_path("sage", big, 256, 5, True, "center", blockdiag=False) == ops.StepPath("layer-calls", "chained+sparse-top")
    assert ops.step_path("sage", big, 256, 5, True, "mean") == ops.StepPath("layer-calls", "chained")               # residue none: single-output lower layers
    assert ops.step_path("sage", big, 64, 5, True, "mean") == ops.StepPath("layer-calls", "chained")
    assert ops.step_path("sage", big, 256, 5, True, "mean", residue="max") == ops.StepPath("layer-calls", "chained")  # dual-output layers chain (round 6)
    assert ops.step_path("sage", big, 64, 5, True, "mean", residue="max") == ops.StepPath("layer-calls", "layer-calls")   # ... at widths in (128, 256] only
    assert ops.step_path("sage", big, 256, 5, True, "center", residue="concat") == ops.StepPath("layer-calls", "layer-calls")
    monkeypatch.setattr(ops, "CHAIN_DUAL", False)
    assert ops.step_path("sage", big, 256, 5, True, "mean", residue="max") == ops.StepPath("layer-calls", "layer-calls")
    monkeypatch.setattr(ops, "CHAIN_DUAL", True)
    assert ops.step_path("sage", big, 256, 5, True, "center", stackable=False) == ops.StepPath("layer-calls", "chained+sparse-top")
    assert ops.step_path("sage", big, 256, 5, False, "center") == ops.StepPath("stack", "none")
    monkeypatch.setattr(ops, "SPARSE_TOP_BWD", False)
    assert ops.step_path("sage", big, 256, 5, True, "center") == ops.StepPath("stack", "stack")
    monkeypatch.setattr(ops, "SAGE_STACK", False)
    assert ops.step_path("sage", big, 256, 5, True, "center") == ops.StepPath("layer-calls", "chained")


def test_gcn_and_gat_rows(monkeypatch):
    assert ops.step_path("gcn", 40000, 256, 3, True, "center") == ops.StepPath("stack", "stack")
    assert ops.step_path("gcn", 40000, 256, 3, True, "mean") == ops.StepPath("layer-calls", "layer-calls")
    big = ops.SPARSE_TOP_BWD_MIN_ROWS
    assert ops.step_path("gat", big, 256, 5, True, "center", heads=4) == ops.StepPath("pair-tail", "rows")
    assert ops.step_path("gat", big - 1, 256, 5, True, "center", heads=4) == ops.StepPath("pair-tail", "dense")
    assert ops.step_path("gat", big, 256, 5, True, "mean", heads=4) == ops.StepPath("pair-tail", "dense")
    assert ops.step_path("gat", big, 256, 5, True, "center", heads=1) == ops.StepPath("node-pass", "rows")          # head width 256 > 128
    assert ops.step_path("gat", big, 64, 5, True, "center", heads=2) == ops.StepPath("node-pass", "rows")
    monkeypatch.setattr(ops, "GAT_PAIR_TAIL", False)
    assert ops.step_path("gat", big, 256, 5, False, "center", heads=4) == ops.StepPath("node-pass", "none")
    with pytest.raises(ValueError):
        ops.step_path("gin", big, 256, 5, True, "center")
