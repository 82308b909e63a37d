"""CPU: the C-ABI library loads and exports every symbol include/zest_render.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "zest_render.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zest_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    import zest_hip
    assert sorted(zest_hip.exported_symbols()) == _declared()


def test_library_exports_every_declared_symbol():
    import zest_hip
    if not os.path.exists(zest_hip.LIB_PATH):
        import build_hip
        build_hip.build()
    lib = ctypes.CDLL(zest_hip.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.zest_abi_version() == 2


def test_missing_library_is_an_error(monkeypatch):
    import zest_hip
    monkeypatch.setattr(zest_hip, "_lib", None)
    monkeypatch.setattr(zest_hip, "LIB_PATH", "/nonexistent/libzest_hip.so")
    with pytest.raises(RuntimeError, match="no non-HIP execution path"):
        zest_hip.lib()


def test_cpu_tensors_are_rejected():
    import torch
    import zest_hip
    with pytest.raises(RuntimeError, match="runs only on a HIP device"):
        zest_hip.embed(torch.zeros(4, 3), 10)


def test_library_is_not_older_than_its_sources():
    """A failed rebuild must not go unnoticed: the shipped .so is at least as new as every source it is
    built from (build_hip.py rebuilds what is stale; __graft_entry__.build() runs it)."""
    import glob
    pkg = os.path.join(ROOT, "zest-nerf_amd")
    lib = os.path.join(pkg, "libzest_hip.so")
    srcs = glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.cuh")) + \
        glob.glob(os.path.join(pkg, "csrc", "*.h")) + [os.path.join(ROOT, "include", "zest_render.h")]
    newest = max(srcs, key=os.path.getmtime)
    assert os.path.getmtime(lib) >= os.path.getmtime(newest), "libzest_hip.so is older than %s: rebuild" % newest
