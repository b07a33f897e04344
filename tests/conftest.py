import os
import sys
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


@pytest.fixture(scope="session")
def hostmath():
    """TEST-ONLY host build of csrc/mht_math.h (tests/hostmath): checks the kernel arithmetic on the CPU."""
    import ctypes
    d = os.path.join(ROOT, "tests", "hostmath")
    so = os.path.join(d, "libhostmath.so")
    src = os.path.join(d, "hostmath.cpp")
    hdrs = [os.path.join(ROOT, "pymht_amd", "csrc", h) for h in ("mht_math.h", "mht_ais_math.h", "mht_la64.h")]
    if (not os.path.exists(so)) or os.path.getmtime(so) < max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in hdrs]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", src, "-o", so])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (pymht_amd has no CPU fallback)")
    from pymht_amd.device import Context
    ctx = Context(0)
    yield ctx
    ctx.close()
