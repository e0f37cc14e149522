import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pmco_py
    pmco_py.build()
    return pmco_py


@pytest.fixture(scope="session")
def wp():
    import wrf_partmc_amd
    wrf_partmc_amd.build_library()
    return wrf_partmc_amd
