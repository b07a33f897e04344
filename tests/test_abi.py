"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/mht_amd.h declares."""
import ctypes
import os

from pymht_amd import _lib


def test_library_loads_and_exports_header_symbols():
    """Both builds of the library -- libmht_amd.so (4 states) and libmht_amd6.so (the same sources with -DMHT_NX=6) -- load and export
    every entry point the header declares; no compute call without a GPU."""
    from pymht_amd import build
    names = _lib.exported_symbols()
    assert "mht_gate_scan" in names and "mht_create" in names and "mht_forest_scan" in names
    for nx in (4, 6):
        lib = _lib.load(nx=nx)
        assert os.path.exists(build.lib_path(nx))
        assert lib.mht_abi_version() == 6
        for name in names:
            assert hasattr(lib, name), "%s does not export %s" % (os.path.basename(build.lib_path(nx)), name)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    from pymht_amd.device import Context
    with pytest.raises(RuntimeError):
        Context(0)


def test_product_does_not_import_oracle():
    root = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "mht_oracle" not in src and "refimport" not in src, f
