#!/usr/bin/env python3
"""Regenerates tests/golden/vectors.json.

The reference (RTL) ships no vectors and cannot be simulated here, so these are outputs of the CPU
oracle (oracle/jpegenc_ref.cpp), each cross-checked at generation time against the independent numpy
restatement (tests/ref_numpy.py).  Inputs are stored as raw RGB bytes (hex) so that no generator has
to stay bit-stable; outputs as hex words plus the end-of-file fields of jpeg_top.
Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
import ref_numpy as N  # noqa: E402
from conftest import natural_like  # noqa: E402


def cases():
    rng = np.random.default_rng(2025)
    y, x = np.mgrid[0:64, 0:64]
    ramp = np.stack([(4 * x) & 255, (4 * y) & 255, (2 * (x + y)) & 255], axis=-1).astype(np.uint8)
    blue = np.zeros((8, 16, 3), np.uint8)
    blue[..., 2] = 255
    sparse_q = [[255] * 64 for _ in range(3)]
    for c in range(3):
        for i in (0, 1, 8, 9, 27, 45):
            sparse_q[c][i] = 2 + c
    yield "flat_grey_64x64", np.full((64, 64, 3), 128, np.uint8), {}
    yield "ramp_64x64", ramp, {}
    yield "random_64x64_seed1", np.random.default_rng(1).integers(0, 256, (64, 64, 3), dtype=np.uint8), {}
    yield "flat_black_8x16", np.zeros((8, 16, 3), np.uint8), {}
    yield "pure_blue_8x16_wrap", blue, {}
    yield "pure_blue_8x16_nowrap", blue, {"rgb_wrap": False}
    yield "natural_24x40", natural_like(24, 40, 11), {}
    yield "binary_noise_16x72", rng.choice(np.array([0, 255], dtype=np.uint8), size=(16, 72, 3)), {}
    yield "random_8x520", rng.integers(0, 256, (8, 520, 3), dtype=np.uint8), {}
    # One image under both settings of the cb_huff 6'd64 switch; the two streams must differ (checked in main).
    # The table keeps zig-zag positions 31 and 51 (entries 35 and 45) and little else, so that Cb blocks have
    # the 16th zero of a run at position 47 with a coefficient still to come: the case the literal decides.
    q3_tab = [[255] * 64 for _ in range(3)]
    for c in range(3):
        for i in (0, 1, 8, 35, 45):
            q3_tab[c][i] = 2 + c
    q3 = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    yield "random_32x32_sparse_quant", q3, {"qtabs": q3_tab}
    yield "random_32x32_sparse_quant_cbfix", q3, {"qtabs": q3_tab, "cb_6d64": False}
    yield "random_32x32_sparse_quant_b", rng.integers(0, 256, (32, 32, 3), dtype=np.uint8), {"qtabs": sparse_q}
    yield "natural_16x16_quant", natural_like(16, 16, 12), {"qtabs": [[3] * 64, [7] * 64, [16] * 64]}


def main():
    out = []
    for name, rgb, kw in cases():
        cfg = O.default_config(qtabs=kw.get("qtabs"), rgb_wrap=kw.get("rgb_wrap", True), cb_6d64=kw.get("cb_6d64", True))
        words, res = O.encode(rgb, cfg)
        ref = N.encode_image(rgb, qtabs=kw.get("qtabs"), wrap_quirk=kw.get("rgb_wrap", True),
                             cb_quirk=kw.get("cb_6d64", True))
        assert ref["words"] == list(words), name
        out.append(dict(name=name, width=int(rgb.shape[1]), height=int(rgb.shape[0]), config=kw,
                        rgb_hex=rgb.tobytes().hex(), words_hex="".join("%08x" % w for w in words),
                        n_full_words=int(res.n_full_words), tail_bits=int(res.tail_bits),
                        tail_word=int(res.tail_word), total_bits_unstuffed=int(res.total_bits_unstuffed),
                        total_bytes_stuffed=int(res.total_bytes_stuffed)))
    by_name = {v["name"]: v for v in out}
    a, b = by_name["random_32x32_sparse_quant"], by_name["random_32x32_sparse_quant_cbfix"]
    assert a["rgb_hex"] == b["rgb_hex"] and a["words_hex"] != b["words_hex"], "the Q3 pair must differ by the switch alone"
    a, b = by_name["pure_blue_8x16_wrap"], by_name["pure_blue_8x16_nowrap"]
    assert a["rgb_hex"] == b["rgb_hex"] and a["words_hex"] != b["words_hex"]
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote %d vectors, %d bytes" % (len(out), os.path.getsize(os.path.join(HERE, "vectors.json"))))


if __name__ == "__main__":
    main()
