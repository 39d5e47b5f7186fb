"""pytest configuration: `gpu` marker, repo root on sys.path, package loader.

`-m "not gpu"` tests run in the CPU-only build container; `-m gpu` tests are the
parity tests proper and need one MI355X.
"""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "object-pose-estimation_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """The package directory has a hyphen, so it is imported through importlib.  A checkout without the built
    library (the .so files are git-ignored) gets it compiled here, exactly as __graft_entry__.build() does."""
    pkg = importlib.import_module(PKG_NAME)
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build_library()
    return pkg


@pytest.fixture(scope="session")
def ope():
    return load_pkg()


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG_NAME + ".synth")
