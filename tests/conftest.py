import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cet-driven-simulation-for-3d-printing-am-kmc-approach_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")

# The product mirrors the reference's flat module layout (import kmc_simulation, ...):
# put its directory on sys.path exactly as a user of the reference would.
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle
