import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _gpu_available()


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle (oracle/*.c) is test infrastructure: build it once per session.  A fresh checkout holds no built libmxv.so
    either (build artefacts are not in the history): build it with hipcc — it cross-compiles without a GPU — so that the
    ABI / no-fallback tests can load it."""
    import subprocess

    from oracle import oracle

    oracle.build()
    lib = os.path.join(ROOT, "gym_amd", "_lib", "libmxv.so")
    if not os.path.exists(lib):
        subprocess.check_call(["bash", os.path.join(ROOT, "gym_amd", "csrc", "build.sh")])
    yield
