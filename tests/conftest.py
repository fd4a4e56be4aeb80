"""pytest configuration: `gpu` marker, import paths, builders for the CPU-side checkers."""
import os
import subprocess
import sys

import pytest

try:  # PyTorch bundles its own HIP runtime: when it shares a process with libmistral_water.so it has to
    import torch  # initialise first (INTEGRATION.md "PyTorch in the same process")
    torch.cuda.is_available()
except Exception:  # torch is optional for everything except the device-pointer tests
    torch = None

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "mistral-water_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))

import memguard  # noqa: E402  host-memory cap for the whole test process (DESIGN.md "Incident")
memguard.install()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """oracle/ -- the CPU restatement of the reference (test infrastructure)."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def emul():
    """tests/emul -- host lock-step emulation of the kernels' phase functions."""
    import emul_build
    return emul_build.load()


@pytest.fixture(scope="session")
def mw():
    """The product: ctypes binding of libmistral_water.so (HIP only)."""
    try:  # when PyTorch shares the process it must initialise ITS bundled HIP runtime first
        import torch
        torch.cuda.is_available()
    except Exception:
        pass
    import mistral_water
    mistral_water.lib()
    from mistral_water import _native
    _native.require_product_build("tests")      # a green test names a product build (MW_ALLOW_LAB=1: an A/B variant under MW_LIB)
    return mistral_water


def has_gpu() -> bool:
    try:
        import mistral_water
        return mistral_water.lib().mw_device_count() > 0
    except Exception:
        return False
