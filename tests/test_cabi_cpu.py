"""CPU-side checks of the product boundary: the C-ABI library loads without a GPU and
exports every symbol include/malva_hip.h declares; with no GPU it refuses to create a
context instead of falling back to anything."""
import os
import re

import pytest

from malva_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "malva_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libmalva_hip.so lacks %s" % n
    assert sorted(capi.EXPORTED) == names


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.MalvaError):
        capi.Context(35, 43, 1 << 20)


def test_product_never_imports_the_oracle():
    """nothing under malva_amd/ may import, include, link or load anything from oracle/"""
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b)|(#include\s*[\"<][^\">]*oracle)|(libmalva_oracle)|(oracle/)")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "malva_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                for n, line in enumerate(open(os.path.join(dirpath, f), errors="replace"), 1):
                    assert not pat.search(line), "%s:%d references the oracle: %s" % (f, n, line.strip())
    mk = open(os.path.join(ROOT, "Makefile")).read()
    lib_rule = mk[mk.index("malva_amd/lib/libmalva_hip.so:"):mk.index("cli:")]
    assert "oracle" not in lib_rule
