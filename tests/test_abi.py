"""CPU: the C-ABI library loads and exports every symbol include/a2v.h declares (no compute)."""
import os
import re

from conftest import ROOT


def header_decls():
    txt = open(os.path.join(ROOT, "include", "a2v.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    decls = {}
    for m in re.finditer(r"\bint\s+(a2v_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_library_exports_every_declared_symbol():
    from animal2vec_amd import _lib
    lib = _lib.load()
    decls = header_decls()
    assert len(decls) >= 25
    for name, nargs in decls.items():
        assert hasattr(lib, name), f"{name} declared in include/a2v.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
        assert len(_lib.SIGNATURES[name]) == nargs, (name, len(_lib.SIGNATURES[name]), nargs)
    assert set(_lib.SIGNATURES) == set(decls)
    assert lib.a2v_abi_version() == 5


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from animal2vec_amd import ops, _lib
    with pytest.raises(_lib.A2VError):
        ops.linear(torch.zeros(8, 8), torch.zeros(8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "animal2vec_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "a2v_oracle" not in src and "ref_shims" not in src and "from oracle" not in src, os.path.join(dp, f)
