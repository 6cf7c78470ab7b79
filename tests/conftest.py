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
