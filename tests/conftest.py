import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def natural_like(h, w, seed):
    """Synthetic 'natural-like' RGB (see the product package's synth module)."""
    import importlib
    return importlib.import_module("jpeg-encoder_amd.synth").natural_like(h, w, seed)


@pytest.fixture(scope="session")
def make_natural():
    return natural_like


def load_pkg():
    """The product package (its directory name is not an identifier, hence importlib)."""
    import importlib
    return importlib.import_module("jpeg-encoder_amd")


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()
