import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    from zeroshotsemanticsegmentation_amd import _lib
    return _lib.load()


@pytest.fixture()
def fast_tmp():
    """scratch directory for the end-to-end CLI tests: they write a full checkpoint (~1.6 GB with the optimizer state) after
    every epoch, exactly like the reference; tmpfs keeps that off the disk when the box has one"""
    import shutil
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="szn_test_", dir=base)
    try:
        yield d
    finally:
        shutil.rmtree(d, ignore_errors=True)
