import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HAS_GPU = torch.cuda.is_available()
if not HAS_GPU:
    # CPU-only session: the kernel sources are exercised through the SIMT emulator (tests/emu/hip_emu.h).
    # This is an explicit opt-in of the test-suite; the product path never sets it.
    os.environ["FSV2V_EMU"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not HAS_GPU:
            item.add_marker(skip_gpu)


@pytest.fixture(scope="session")
def emu_lib():
    """Build (if stale) and select the emulated kernel library for CPU logic tests."""
    if HAS_GPU:
        pytest.skip("emulated-kernel tests run in the CPU-only session")
    import importlib
    import fsv2v_amd  # noqa: F401
    build = importlib.import_module("few-shot-vid2vid_amd.build")
    build.build_emu()
    return importlib.import_module("few-shot-vid2vid_amd.lib")


@pytest.fixture(scope="session")
def hip_lib():
    import importlib
    import fsv2v_amd  # noqa: F401
    lib = importlib.import_module("few-shot-vid2vid_amd.lib")
    lib.get_lib()           # raises loudly when libfsv2v_hip.so is missing
    assert not lib.is_emu()
    return lib
