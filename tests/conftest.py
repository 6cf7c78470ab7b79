"""pytest configuration: registers the `gpu` marker and builds the CPU oracle on demand.

`-m "not gpu"` runs everything that needs no device (oracle KATs, host logic, C-ABI symbol
checks); `-m gpu` runs the parity tests proper on an MI355X through the C ABI.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s of CPU")


def _hip_devices():
    try:
        import ddo_amd

        return ddo_amd.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without an MI355X skips the gpu-marked tests instead of drowning real failures
    in device errors.  An explicit `-m gpu` selection (the GPU box) or DDO_REQUIRE_GPU=1 keeps them strict: there the
    tests FAIL when no device is visible -- a silent skip would read as a pass."""
    strict = os.environ.get("DDO_REQUIRE_GPU") == "1" or "gpu" in (config.getoption("-m") or "").replace("not gpu", "")
    if strict or _hip_devices() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible (gpu-marked tests run on the MI355X box: pytest -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_build():
    """Path of oracle/_build (built here with make when missing or stale)."""
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["make", "-s", "-C", odir], check=True)
    return os.path.join(odir, "_build")


@pytest.fixture(scope="session")
def oracle(oracle_build):
    from tests.oracle_binding import Oracle

    return Oracle(os.path.join(oracle_build, "liboracle.so"))


def data_path(*parts):
    return os.path.join(ROOT, "data", *parts)
