"""CPU: the C-ABI library builds, loads, exports every symbol include/graal_hip.h declares, and refuses to run
without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "graal_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(graal_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    from graal_amd import lib
    assert declared_symbols() == sorted(lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from graal_amd import build as gbuild
    L = ctypes.CDLL(gbuild.build_hip())
    for sym in declared_symbols():
        assert hasattr(L, sym), sym
    L.graal_abi_version.restype = ctypes.c_int
    from graal_amd import lib
    assert L.graal_abi_version() == lib.ABI_VERSION == 7


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from graal_amd.lib import Engine, GraalError
    with pytest.raises(GraalError, match="no HIP device|hip"):
        Engine(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "graal_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libgraal_oracle" not in src, f


def test_environment_switches_are_documented():
    """Every GRAAL_* environment variable the library, the host package or bench.py reads is listed in INTEGRATION.md section 5
    (tuning knobs in the product library must not drift undocumented)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    used = set()
    for f in glob.glob(os.path.join(root, "graal_amd", "csrc", "*")) + glob.glob(os.path.join(root, "graal_amd", "*.py")) + [os.path.join(root, "bench.py")]:
        if os.path.isfile(f) and not f.endswith(".so"):
            txt = open(f, errors="ignore").read()
            used |= set(re.findall(r'getenv\("(GRAAL_[A-Z0-9_]+)"\)', txt))
            used |= set(re.findall(r'environ(?:\.get)?[\[(]\s*"(GRAAL_[A-Z0-9_]+)"', txt))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    documented = set(re.findall(r"`(GRAAL_[A-Z0-9_]+)`", doc))
    assert used, "no switches found: the patterns of this test are out of date"
    assert used <= documented, sorted(used - documented)
