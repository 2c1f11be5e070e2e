"""Committed golden vectors (tests/golden/vectors.json): the oracle must keep producing them (CPU), and
the HIP path must produce them too (GPU).  The vectors are oracle outputs cross-checked against the
numpy restatement when they were generated; the reference itself ships none (parity unpinned)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import load_pkg

VEC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.json")))


def unpack(v):
    rgb = np.frombuffer(bytes.fromhex(v["rgb_hex"]), dtype=np.uint8).reshape(v["height"], v["width"], 3)
    words = np.array([int(v["words_hex"][i:i + 8], 16) for i in range(0, len(v["words_hex"]), 8)], dtype=np.uint32)
    return rgb, words


@pytest.mark.parametrize("v", VEC, ids=[v["name"] for v in VEC])
def test_oracle_reproduces_golden(v):
    rgb, want = unpack(v)
    kw = v["config"]
    cfg = O.default_config(qtabs=kw.get("qtabs"), rgb_wrap=kw.get("rgb_wrap", True), cb_6d64=kw.get("cb_6d64", True))
    words, res = O.encode(rgb, cfg)
    assert np.array_equal(words, want)
    assert (res.n_full_words, res.tail_bits, res.tail_word) == (v["n_full_words"], v["tail_bits"], v["tail_word"])
    assert res.total_bits_unstuffed == v["total_bits_unstuffed"]
    assert res.total_bytes_stuffed == v["total_bytes_stuffed"]


@pytest.mark.gpu
@pytest.mark.parametrize("v", VEC, ids=[v["name"] for v in VEC])
def test_hip_path_reproduces_golden(v):
    pkg = load_pkg()
    rgb, want = unpack(v)
    kw = v["config"]
    with pkg.JpegTop(qtabs=kw.get("qtabs"), rgb_wrap=kw.get("rgb_wrap", True), cb_6d64=kw.get("cb_6d64", True)) as e:
        words, res = e.encode(rgb)
    assert np.array_equal(words, want)
    assert (int(res["n_full_words"]), int(res["tail_bits"]), int(res["tail_word"])) == \
           (v["n_full_words"], v["tail_bits"], v["tail_word"])
    assert int(res["total_bits_unstuffed"]) == v["total_bits_unstuffed"]
    assert int(res["total_bytes_stuffed"]) == v["total_bytes_stuffed"]
