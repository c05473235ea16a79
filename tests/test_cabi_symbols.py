"""CPU: the C-ABI library loads and exports every symbol include/shadow_hip.h declares."""
import os
import re

from shadow_gnn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if not fn.endswith(".h"):
            continue
        txt = open(os.path.join(ROOT, "include", fn)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for m in re.finditer(r"\b((?:sg|sl)_[a-z0-9_]+)\s*\(", txt):
            names.add(m.group(1))
    return names


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in shadow_gnn_amd/_lib.py"
    for name in _lib.SIGNATURES:
        assert name in declared, f"{name} bound in _lib.py but not declared in include/"


def test_abi_version_and_error_string():
    lib = _lib.load()
    assert lib.sg_abi_version() >= 1
    assert isinstance(lib.sg_last_error(), bytes)
