"""pytest setup: marker registration and import paths.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol check (no compute).
`-m gpu`      : parity tests proper -- HIP path (through the C-ABI) vs the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-stylization-with-nca_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
