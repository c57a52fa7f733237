"""Shared fixtures.  GPU tests are marked `gpu`; everything else runs on a machine without one."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def orc():
    return graft.load_oracle()


@pytest.fixture(scope="session")
def ctx(pkg):
    """One GPU context for the whole session (GPU tests only)."""
    c = pkg.Context(0)
    yield c
    c.close()
