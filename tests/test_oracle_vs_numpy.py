"""Cross-check the C++ oracle against the independent numpy restatement on random, natural-like
and adversarial inputs (the reference has no vectors of its own: parity is otherwise unpinned)."""
import numpy as np
import pytest

import oracle_lib as O
import ref_numpy as N
from conftest import natural_like


def compare(rgb, qtabs=None, **quirks):
    cfg = O.default_config(qtabs=qtabs, rgb_wrap=quirks.get("wrap_quirk", True), cb_6d64=quirks.get("cb_quirk", True))
    words, res = O.encode(rgb, cfg)
    ref = N.encode_image(rgb, qtabs=qtabs, **quirks)
    assert res.total_bits_unstuffed == ref["total_bits"]
    assert res.total_bytes_stuffed == ref["stuffed_bytes"]
    assert (res.n_full_words, res.tail_bits, res.tail_word) == (ref["n_full_words"], ref["tail_bits"], ref["tail_word"])
    assert list(words) == ref["words"]
    return res


def test_config1_images_64x64():
    y, x = np.mgrid[0:64, 0:64]
    ramp = np.stack([(4 * x) & 255, (4 * y) & 255, (2 * (x + y)) & 255], axis=-1).astype(np.uint8)
    compare(np.full((64, 64, 3), 128, np.uint8))
    compare(ramp)
    compare(np.random.default_rng(1).integers(0, 256, (64, 64, 3), dtype=np.uint8))


@pytest.mark.parametrize("seed", range(4))
def test_random_small(seed):
    rng = np.random.default_rng(100 + seed)
    h, w = [(8, 8), (8, 40), (24, 16), (32, 32)][seed]
    compare(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))


def test_natural_like_and_coefficients():
    rgb = natural_like(48, 64, 2)
    compare(rgb)
    got = O.image_coeffs(rgb)
    want = N.image_coeffs(rgb)
    assert np.array_equal(got.astype(np.int64), want)


def test_extremes_and_saturated_colours():
    rng = np.random.default_rng(7)
    rgb = rng.choice(np.array([0, 255], dtype=np.uint8), size=(32, 32, 3))
    compare(rgb)
    rgb = np.zeros((16, 16, 3), np.uint8)
    rgb[..., 2] = 255  # pure blue: Cb wraps (Q1)
    compare(rgb)
    compare(rgb, wrap_quirk=False)
    rgb = np.zeros((16, 16, 3), np.uint8)
    rgb[..., 0] = 255  # pure red: Cr wraps
    compare(rgb)
    # checkerboards at pixel and block granularity
    y, x = np.mgrid[0:32, 0:32]
    compare((((x + y) & 1) * 255).astype(np.uint8)[..., None].repeat(3, -1))
    compare(((((x >> 3) + (y >> 3)) & 1) * 255).astype(np.uint8)[..., None].repeat(3, -1))


def test_nondefault_quant_tables():
    rng = np.random.default_rng(9)
    rgb = natural_like(32, 32, 3)
    qt = [rng.integers(1, 64, 64), rng.integers(1, 100, 64), np.full(64, 16)]
    qt[0][5] = 0  # Q == 0 -> multiplier 0
    res = compare(rgb, qtabs=qt)
    assert res.total_bits_unstuffed < O.encode(rgb)[1].total_bits_unstuffed
    compare(rng.integers(0, 256, (16, 24, 3), dtype=np.uint8), qtabs=qt, cb_quirk=False)


def test_sparse_blocks_hit_zrl_rules():
    # quantise hard so that long zero runs (and the Q2/Q3 positions) occur in real pipelines
    rng = np.random.default_rng(21)
    rgb = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    qt = [np.full(64, 255)] * 3
    for trial in range(3):
        q = [t.copy() for t in qt]
        for c in range(3):
            idx = rng.integers(0, 64, 6)
            q[c][idx] = rng.integers(1, 8, 6)
        compare(rgb, qtabs=q)
        compare(rgb, qtabs=q, cb_quirk=False)


def test_ff_bytes_occur_and_are_stuffed():
    rgb = np.random.default_rng(3).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    words, res = O.encode(rgb)
    raw_bytes = (res.total_bits_unstuffed // 8)
    assert res.total_bytes_stuffed > raw_bytes  # at least one 0xFF byte was stuffed
    by = words.astype(">u4").tobytes()[:res.total_bytes_stuffed]
    for i in range(len(by) - 1):
        if by[i] == 0xFF:
            assert by[i + 1] == 0x00
