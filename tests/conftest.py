import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _hip_device_available() -> bool:
    try:
        from caliscope_amd import build
        from caliscope_amd.hip_engine import device_count

        build.build(verbose=False)
        return device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """Plain ``pytest tests`` on a host without a HIP device: the ``gpu`` tests are skipped instead of failing in their
    fixtures (``-m gpu`` on the GPU box still fails loudly if the library or the device is missing: nothing is skipped there)."""
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items or _hip_device_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (run with -m gpu on an MI355X box)")
    for it in gpu_items:
        it.add_marker(skip)
