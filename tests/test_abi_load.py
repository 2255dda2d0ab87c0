"""CPU-side checks of the C-ABI library: it loads and exports every symbol the header declares.
No compute call is made here (there is no GPU in the build container)."""
import os
import re


def test_library_exports_every_declared_symbol(pkg):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "ascii_rast.h")).read()
    declared = set(re.findall(r"\b(ar_[a-z_]+)\s*\(", header))
    assert declared == set(pkg.binding.EXPORTS), declared ^ set(pkg.binding.EXPORTS)
    lib = pkg.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg.binding, "_lib", None)
    monkeypatch.setattr(pkg.binding, "LIB_PATH", "/nonexistent/libascii_rast.so")
    try:
        pkg.binding.load_library()
    except pkg.AsciiRastError as e:
        assert "not built" in str(e)
    else:
        raise AssertionError("loading a missing library must raise")
