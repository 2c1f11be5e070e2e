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
    """Synthetic 'natural-like' RGB: sum of 6 octaves of value noise per channel, gamma 2.2, 8 bit."""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w, 3), dtype=np.float64)
    amp_total = 0.0
    for octave in range(6):
        cells = 2 ** (octave + 2)
        gh, gw = cells + 2, cells + 2
        grid = rng.random((gh, gw, 3))
        ys = np.linspace(0, cells, h, endpoint=False)
        xs = np.linspace(0, cells, w, endpoint=False)
        y0 = ys.astype(int)
        x0 = xs.astype(int)
        fy = (ys - y0)[:, None, None]
        fx = (xs - x0)[None, :, None]
        fy = fy * fy * (3 - 2 * fy)
        fx = fx * fx * (3 - 2 * fx)
        g00 = grid[y0][:, x0]
        g01 = grid[y0][:, x0 + 1]
        g10 = grid[y0 + 1][:, x0]
        g11 = grid[y0 + 1][:, x0 + 1]
        layer = (g00 * (1 - fx) + g01 * fx) * (1 - fy) + (g10 * (1 - fx) + g11 * fx) * fy
        amp = 0.5 ** octave
        img += amp * layer
        amp_total += amp
    img /= amp_total
    img = np.clip(img, 0, 1) ** (1 / 2.2)
    return np.round(img * 255).astype(np.uint8)


@pytest.fixture(scope="session")
def make_natural():
    return natural_like
