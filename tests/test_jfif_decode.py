"""The oracle's stream, wrapped in JFIF headers, must decode with an independent JPEG decoder (libjpeg
through Pillow) to nearly the original image: at the RTL's all-ones quantisation the codec is almost
lossless.  This pins the entropy coder, zig-zag, table pairing (K.4 DC + K.5 AC for all components),
DCT orientation/scaling and the interleave order against something outside this repository."""
import io

import numpy as np
import pytest

import oracle_lib as O
from conftest import load_pkg, natural_like

Image = pytest.importorskip("PIL.Image")


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def decode(jpg):
    return np.asarray(Image.open(io.BytesIO(jpg)).convert("RGB"))


def test_default_tables_decode_nearly_lossless():
    pkg = load_pkg()
    rgb = natural_like(64, 96, 4)
    words, res = O.encode(rgb)
    jpg = pkg.jfif.wrap_jfif(words, res.n_full_words, res.tail_bits, 96, 64)
    out = decode(jpg)
    assert out.shape == rgb.shape
    assert psnr(out, rgb) > 40.0


def test_asymmetric_content_orientation():
    # strong horizontal vs vertical structure: a transposed DCT or zig-zag would decode to garbage
    pkg = load_pkg()
    y, x = np.mgrid[0:32, 0:64]
    rgb = np.stack([(x * 4) & 255, (y * 8) & 255, ((x // 8) * 32) & 255], axis=-1).astype(np.uint8)
    words, res = O.encode(rgb)
    out = decode(pkg.jfif.wrap_jfif(words, res.n_full_words, res.tail_bits, 64, 32))
    assert psnr(out, rgb) > 35.0


def test_quantised_stream_decodes():
    pkg = load_pkg()
    rgb = natural_like(64, 64, 6)
    qt = [np.full(64, 8), np.full(64, 16), np.full(64, 16)]  # powers of two: the reciprocal is exact
    words, res = O.encode(rgb, O.default_config(qtabs=qt))
    out = decode(pkg.jfif.wrap_jfif(words, res.n_full_words, res.tail_bits, 64, 64, qtabs=qt))
    assert psnr(out, rgb) > 28.0


def test_noise_decodes():
    pkg = load_pkg()
    rgb = np.random.default_rng(8).integers(0, 256, (32, 32, 3), dtype=np.uint8)
    words, res = O.encode(rgb)
    out = decode(pkg.jfif.wrap_jfif(words, res.n_full_words, res.tail_bits, 32, 32))
    assert psnr(out, rgb) > 30.0


def test_asymmetric_tables_in_both_index_orders():
    """A table that is coarse for high horizontal frequencies only.  Written in the conventional row-major
    (vertical, horizontal) order and passed with quant_index_order = 1 it must give the stream of its
    transpose under the default order, and an external decoder fed the same table must reproduce the image:
    that pins which table entry meets which frequency."""
    pkg = load_pkg()
    rgb = natural_like(64, 64, 9)
    conv = np.array([[32 if b >= 3 else 2 for b in range(8)] for a in range(8)]).reshape(64)  # [a*8+b]
    as_hooked = conv.reshape(8, 8).T.reshape(64)                                            # [i*8+j], i horizontal
    w1, r1 = O.encode(rgb, O.default_config(qtabs=[conv] * 3, quant_index_order=1))
    w0, r0 = O.encode(rgb, O.default_config(qtabs=[as_hooked] * 3))
    assert np.array_equal(w0, w1) and r0.tail_bits == r1.tail_bits
    wrong, _ = O.encode(rgb, O.default_config(qtabs=[conv] * 3))  # the same numbers read the other way round
    assert not np.array_equal(wrong[:64], w1[:64])
    good = decode(pkg.jfif.wrap_jfif(w1, r1.n_full_words, r1.tail_bits, 64, 64, qtabs=[conv] * 3, quant_index_order=1))
    also = decode(pkg.jfif.wrap_jfif(w0, r0.n_full_words, r0.tail_bits, 64, 64, qtabs=[as_hooked] * 3))
    bad = decode(pkg.jfif.wrap_jfif(w1, r1.n_full_words, r1.tail_bits, 64, 64, qtabs=[as_hooked] * 3, quant_index_order=1))
    assert np.array_equal(good, also)
    assert psnr(good, rgb) > 28.0 and psnr(good, rgb) > psnr(bad, rgb) + 3.0


def test_padded_stream_in_a_container_with_the_true_size():
    """The edge-replication extension codes the padded image; the JFIF header carries the true size and the
    decoder crops."""
    pkg = load_pkg()
    rgb = natural_like(64, 64, 10)[:53, :37]
    padded = np.pad(rgb, ((0, 3), (0, 3), (0, 0)), mode="edge")
    words, res = O.encode(padded)
    out = decode(pkg.jfif.wrap_jfif(words, res.n_full_words, res.tail_bits, 37, 53))
    assert out.shape == rgb.shape and psnr(out, rgb) > 40.0
