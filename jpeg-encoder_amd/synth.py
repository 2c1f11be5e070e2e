"""Synthetic inputs for tests and benchmarks (nothing here is on the product path).

`natural_like`: sum of 6 octaves of value noise per channel (cell sizes 64, 32, ..., 2 pixels,
amplitude x0.7 per octave), gamma 2.2, 8 bit.  At the RTL's default all-ones quantisation it codes to
about 1 byte per pixel, the regime of photographic content; uniform noise codes to about 4.2.
"""
import numpy as np

OCTAVES = 6
BASE_CELL = 64
PERSISTENCE = 0.7


def natural_like(h, w, seed):
    """(h, w, 3) uint8 numpy image."""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w, 3), dtype=np.float32)
    amp_total = 0.0
    for octave in range(OCTAVES):
        cell = max(1, BASE_CELL >> octave)
        grid = rng.random((h // cell + 3, w // cell + 3, 3), dtype=np.float32)
        ys = np.arange(h, dtype=np.float32) / cell
        xs = np.arange(w, dtype=np.float32) / cell
        y0 = ys.astype(np.int32)
        x0 = xs.astype(np.int32)
        fy = (ys - y0)[:, None, None]
        fx = (xs - x0)[None, :, None]
        fy = fy * fy * (3 - 2 * fy)
        fx = fx * fx * (3 - 2 * fx)
        top = grid[y0][:, x0] * (1 - fx) + grid[y0][:, x0 + 1] * fx
        bot = grid[y0 + 1][:, x0] * (1 - fx) + grid[y0 + 1][:, x0 + 1] * fx
        amp = PERSISTENCE ** octave
        img += amp * (top * (1 - fy) + bot * fy)
        amp_total += amp
    img /= amp_total
    img = np.clip(img, 0, 1) ** (1 / 2.2)
    return np.round(img * 255).astype(np.uint8)


def natural_like_torch(n, h, w, seed, device):
    """(n, h, w, 3) uint8 torch tensor generated on `device` (same recipe, torch RNG)."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    img = torch.zeros((n, h, w, 3), dtype=torch.float32, device=device)
    amp_total = 0.0
    for octave in range(OCTAVES):
        cell = max(1, BASE_CELL >> octave)
        grid = torch.rand((n, h // cell + 3, w // cell + 3, 3), generator=gen, device=device, dtype=torch.float32)
        ys = torch.arange(h, device=device, dtype=torch.float32) / cell
        xs = torch.arange(w, device=device, dtype=torch.float32) / cell
        y0 = ys.floor().long()
        x0 = xs.floor().long()
        fy = (ys - y0)[None, :, None, None]
        fx = (xs - x0)[None, None, :, None]
        fy = fy * fy * (3 - 2 * fy)
        fx = fx * fx * (3 - 2 * fx)
        rows0 = grid[:, y0]
        rows1 = grid[:, y0 + 1]
        top = rows0[:, :, x0] * (1 - fx) + rows0[:, :, x0 + 1] * fx
        bot = rows1[:, :, x0] * (1 - fx) + rows1[:, :, x0 + 1] * fx
        amp = PERSISTENCE ** octave
        img += amp * (top * (1 - fy) + bot * fy)
        amp_total += amp
        del grid, rows0, rows1, top, bot
    img /= amp_total
    img = img.clamp_(0, 1).pow_(1 / 2.2).mul_(255).round_()
    return img.to(torch.uint8)
