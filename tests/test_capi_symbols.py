"""CPU-only: the C-ABI library loads and exports every symbol that include/enoki_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "enoki_hip.h")).read()
    return sorted(set(re.findall(r"EK_API\s+[\w\s\*]+?\b(ek_hip_\w+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    assert len(syms) >= 38
    for s in ("ek_hip_malloc", "ek_hip_unary", "ek_hip_gather", "ek_hip_scatter_add", "ek_hip_reduce", "ek_hip_hsum_safe_mul"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "enoki_amd", "libenoki-hip.so"))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_lists_the_same_symbols():
    from enoki_amd import capi
    assert sorted(capi.EXPORTS) == declared_symbols()


def test_autodiff_library_loads():
    lib = ctypes.CDLL(os.path.join(ROOT, "enoki_amd", "libenoki-hip-autodiff.so"))
    assert lib is not None


def test_no_product_file_references_the_oracle():
    """the product must never import / link / call anything under oracle/"""
    bad = []
    for base in ("enoki_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".cpp", ".hip")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if f == "_build.py":
                        continue     # builds the checkers, does not use them
                    if re.search(r"oracle[/_.]|orc_|libenoki_oracle|libenoki_ref", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
