"""The Huffman coders of the HIP path on directed coefficient patterns, through jpegenc_encode_coeffs.

Pixels cannot steer the coders into their corners (a zero run whose 16th zero lands on position 30 or 46 in front
of an otherwise empty tail, a last coefficient that is not zero, a DC difference of 12 bits, the AC value -1024), so
these tests put coefficients straight on the coders' inputs.  Expected streams: written out by hand from the code
tables and the rules of rtl/4_huffman/cr_huff.sv:213-231 (end_of_block), :558 (ZRL), :518/:656 (last slot),
:1411-1486 (size on bits [9:0]), cb_huff.sv:238-241 (the 6'd64 literal), and, for the fuzz, the CPU oracle fed
the same coefficients.  Bit-exact, every word, tail word and count.
"""
import numpy as np
import pytest

import oracle_lib as O
import ref_numpy as N
from conftest import load_pkg, natural_like

pytestmark = pytest.mark.gpu

ZRL = "11111111001"
EOB = "1010"
EMPTY = "00" + EOB  # a block of zeros behind a predictor equal to its DC


def ac(sym):
    code, length = N.AC_TAB[sym]
    return format(code, "0%db" % length)


@pytest.fixture(scope="module", params=["joined", "single"])
def enc(request):
    """The coder that joins symbols before appending them (small LDS regions, the default for photographs) and the
    one that appends symbol by symbol (large regions); a chunk that overflows its regions runs the third, guarded
    instantiation in either."""
    pkg = load_pkg()
    e = pkg.JpegTop()
    e.set_region_mode(e.REGIONS_LARGE if request.param == "single" else e.REGIONS_SMALL)
    yield e
    e.close()


def words_of_bits(bits):
    r = N.pack_stream(int(bits, 2) if bits else 0, len(bits))
    return np.array(r["words"], dtype=np.uint32), r


def one_mcu(z, comp):
    c = np.zeros((1, 3, 64), dtype=np.int16)
    c[0, comp] = z
    return c


def check_hand(enc, z, comp, block_bits):
    """One MCU whose component `comp` holds z (reading order); the other two blocks are empty."""
    parts = [EMPTY, EMPTY, EMPTY]
    parts[comp] = block_bits
    want, r = words_of_bits("".join(parts))
    words, res = enc.encode_coeffs(one_mcu(z, comp), 8, 8)
    assert int(res["status"]) == 0
    assert int(res["total_bits_unstuffed"]) == sum(len(p) for p in parts)
    assert (int(res["n_full_words"]), int(res["tail_bits"])) == (r["n_full_words"], r["tail_bits"])
    assert np.array_equal(words, want), (list(map(hex, words)), list(map(hex, want)))
    # and the oracle agrees with the hand-written answer
    ow, _ = O.encode_coeffs(one_mcu(z, comp))
    assert np.array_equal(ow, want)


@pytest.mark.parametrize("comp", [0, 1, 2])
def test_runs_and_the_coarse_end_of_block_rule(enc, comp):
    z = [0] * 64
    z[1], z[5] = 1, -2
    check_hand(enc, z, comp, "00" + ac(0x01) + "1" + ac(0x32) + "01" + EOB)
    z = [0] * 64
    z[20] = 3  # 19 zeros before it: the 16th at position 16 emits a ZRL
    check_hand(enc, z, comp, "00" + ZRL + ac(0x32) + "11" + EOB)
    # the 16th zero on position 30, z[31] != 0, nothing behind: end_of_block is already high there, no ZRL,
    # and the coefficient goes out with run 0
    z = [0] * 64
    z[14], z[31] = 1, 1
    check_hand(enc, z, comp, "00" + ac(0xD1) + "1" + ac(0x01) + "1" + EOB)
    z[40] = 1  # with anything behind position 31 the ZRL is there
    check_hand(enc, z, comp, "00" + ac(0xD1) + "1" + ZRL + ac(0x01) + "1" + ac(0x81) + "1" + EOB)
    # a trailing run never emits, wherever it starts
    for last in (1, 13, 14, 15, 29, 30, 31, 45, 46, 47):
        z = [0] * 64
        z[last] = 1
        check_hand(enc, z, comp, "00" + ZRL * ((last - 1) // 16) + ac((((last - 1) % 16) << 4) | 1) + "1" + EOB)


def test_position_46_and_the_6d64_literal(enc):
    z = [0] * 64
    z[30], z[47] = 1, 1  # the 16th zero behind z[30] lands on position 46, z[48..63] empty
    for comp in (0, 1, 2):
        check_hand(enc, z, comp, "00" + ZRL + ac(0xD1) + "1" + ac(0x01) + "1" + EOB)
    z[50] = 1  # a later coefficient: Y and Cr emit the ZRL, Cb (6'd64 reads as 0) still drops it
    for comp in (0, 2):
        check_hand(enc, z, comp, "00" + ZRL + ac(0xD1) + "1" + ZRL + ac(0x01) + "1" + ac(0x21) + "1" + EOB)
    check_hand(enc, z, 1, "00" + ZRL + ac(0xD1) + "1" + ac(0x01) + "1" + ac(0x21) + "1" + EOB)
    # the literal repaired: Cb as Y and Cr
    pkg = load_pkg()
    with pkg.JpegTop(cb_6d64=False) as e:
        words, _ = e.encode_coeffs(one_mcu(z, 1), 8, 8)
    want, _ = words_of_bits(EMPTY + "00" + ZRL + ac(0xD1) + "1" + ZRL + ac(0x01) + "1" + ac(0x21) + "1" + EOB + EMPTY)
    assert np.array_equal(words, want)


@pytest.mark.parametrize("comp", [0, 1, 2])
def test_last_coefficient_and_minus_1024(enc, comp):
    z = [0] * 64
    z[63], z[60] = 5, -1  # 59 zeros: ZRLs at 16, 32, 48 (Cb drops the one at 48); no EOB behind a non-zero z[63]
    check_hand(enc, z, comp, "00" + ZRL * (2 if comp == 1 else 3) + ac(0xB1) + "0" + ac(0x23) + "101")
    # -1024: no clear bit among (v - 1)[9:0], so size 0 and no amplitude; (run 0, size 0) is EOB's code, the map
    # sends (run 1..14, size 0) to the table's first entry '00', (run 15, size 0) is ZRL's code
    z = [0] * 64
    z[1] = -1024
    check_hand(enc, z, comp, "00" + EOB + EOB)
    z = [0] * 64
    z[3] = -1024
    check_hand(enc, z, comp, "00" + "00" + EOB)
    z = [0] * 64
    z[16] = -1024  # 15 zeros before it: the run counter holds 15 and the value itself takes the (15, 0) entry
    check_hand(enc, z, comp, "00" + ZRL + EOB)
    z = [0] * 64
    z[63] = -1024  # the last slot, run 14 by then (62 zeros modulo 16): ZRLs at 16, 32, 48, then '00', no EOB
    check_hand(enc, z, comp, "00" + ZRL * (2 if comp == 1 else 3) + "00")


def test_dc_differences_of_twelve_bits(enc):
    # two MCUs: the second block's difference against the first
    for prev, cur in ((0, 1023), (1023, -1024), (-1024, 1023), (5, 5), (0, -1), (0, 1), (-1024, -1024)):
        c = np.zeros((2, 3, 64), dtype=np.int16)
        c[0, :, 0] = prev
        c[1, :, 0] = cur
        words, res = enc.encode_coeffs(c, 16, 8)
        want, r = O.encode_coeffs(c)
        assert np.array_equal(words, want) and int(res["tail_bits"]) == r.tail_bits
    c = np.zeros((2, 3, 64), dtype=np.int16)
    c[0, 0, 0], c[1, 0, 0] = -1024, 1023  # d = 2047: size 11, amplitude eleven ones
    first = "11111111110" + "01111111111" + EOB
    second = "11111111110" + "11111111111" + EOB
    want, _ = words_of_bits(first + EMPTY + EMPTY + second + EMPTY + EMPTY)
    words, _ = enc.encode_coeffs(c, 16, 8)
    assert np.array_equal(words, want)


def directed_blocks():
    """Every pattern above plus neighbours of the two positions the coarse rule gets wrong."""
    out = []
    for first in range(0, 48):
        for gap in (15, 16, 17, 31, 32, 33):
            k = first + gap + 1
            if k > 63:
                continue
            for behind in (None, k + 1, 40, 50, 63):
                z = np.zeros(64, dtype=np.int16)
                if first:
                    z[first] = 1
                z[k] = -3
                if behind is not None and behind > k and behind <= 63:
                    z[behind] = 7
                out.append(z)
    for k in range(1, 64):
        z = np.zeros(64, dtype=np.int16)
        z[k] = -1024
        out.append(z)
        z = z.copy()
        z[(k + 17) % 63 + 1] = 1023
        out.append(z)
    return out


def test_directed_patterns_in_every_component_and_lane(enc):
    blocks = directed_blocks()
    rng = np.random.default_rng(5)
    n_mcu = 16 * 9 * 4  # 9 chunks: the patterns meet every lane position, chunk seams and the predecessor's DC
    c = np.zeros((n_mcu, 3, 64), dtype=np.int16)
    order = rng.permutation(len(blocks) * 3)
    for i, j in enumerate(order[: n_mcu * 3]):
        c[i // 3, i % 3] = blocks[j % len(blocks)]
    c[:, :, 0] = rng.integers(-1024, 1024, (n_mcu, 3))
    words, res = enc.encode_coeffs(c, 16 * 8 * 2, 9 * 8 * 2)
    want, r = O.encode_coeffs(c)
    assert int(res["total_bits_unstuffed"]) == r.total_bits_unstuffed
    assert (int(res["n_full_words"]), int(res["tail_bits"]), int(res["tail_word"])) == (r.n_full_words, r.tail_bits, r.tail_word)
    assert np.array_equal(words, want)


@pytest.mark.parametrize("kind", ["sparse", "dense", "extremes", "all_minus_1024", "all_1023"])
def test_random_coefficients_against_the_oracle(enc, kind):
    rng = np.random.default_rng({"sparse": 1, "dense": 2, "extremes": 3, "all_minus_1024": 4, "all_1023": 5}[kind])
    n_mcu = 200  # 64 + 64 + 64 + 8: a ragged last chunk
    if kind == "sparse":
        c = rng.integers(-1024, 1024, (n_mcu, 3, 64)) * (rng.random((n_mcu, 3, 64)) < 0.06)
    elif kind == "dense":  # every lane beyond any region: the guarded exact-fit passes
        c = rng.integers(-1024, 1024, (n_mcu, 3, 64))
    elif kind == "extremes":
        c = rng.choice(np.array([-1024, -1023, -512, -1, 0, 0, 0, 1, 511, 1023]), (n_mcu, 3, 64))
    elif kind == "all_minus_1024":
        c = np.full((n_mcu, 3, 64), -1024)
    else:
        c = np.full((n_mcu, 3, 64), 1023)
        c[::2] = -1023  # DC differences of +-2046 between neighbours
    c = c.astype(np.int16)
    words, res = enc.encode_coeffs(c, 200, 64)  # 25 x 8 MCUs
    want, r = O.encode_coeffs(c)
    assert int(res["status"]) == 0
    assert int(res["total_bits_unstuffed"]) == r.total_bits_unstuffed
    assert int(res["total_bytes_stuffed"]) == r.total_bytes_stuffed
    assert (int(res["n_full_words"]), int(res["tail_bits"]), int(res["tail_word"])) == (r.n_full_words, r.tail_bits, r.tail_word)
    assert np.array_equal(words, want)


def test_values_are_taken_modulo_the_ports_eleven_bits(enc):
    rng = np.random.default_rng(9)
    c = (rng.integers(-1024, 1024, (70, 3, 64)) * (rng.random((70, 3, 64)) < 0.2)).astype(np.int16)
    wide = (c + 2048 * rng.integers(-8, 8, c.shape)).astype(np.int16)  # same low 11 bits
    a, _ = enc.encode_coeffs(c, 80, 56)
    b, _ = enc.encode_coeffs(wide, 80, 56)
    assert np.array_equal(a, b) and np.array_equal(a, O.encode_coeffs(wide)[0])


def test_probe_out_and_probe_in_give_the_images_stream(enc):
    """jpegenc_image_coeffs followed by jpegenc_encode_coeffs is jpegenc_encode_rgb8."""
    for rgb in (natural_like(72, 136, 3), np.random.default_rng(2).integers(0, 256, (64, 64, 3), dtype=np.uint8)):
        h, w, _ = rgb.shape
        words, res = enc.encode(rgb)
        coeffs = enc.image_coeffs(rgb)
        again, res2 = enc.encode_coeffs(coeffs, w, h)
        assert np.array_equal(words, again)
        assert int(res["total_bits_unstuffed"]) == int(res2["total_bits_unstuffed"])
        assert int(res["tail_word"]) == int(res2["tail_word"])


def test_bad_arguments(enc):
    pkg = load_pkg()
    c = np.zeros((1, 3, 64), dtype=np.int16)
    words = np.zeros(64, dtype=np.uint32)
    res = np.zeros(1, dtype=pkg.encoder.RESULT_DTYPE)
    L = enc.lib
    assert L.jpegenc_encode_coeffs(enc._ctx, c.ctypes.data, 12, 8, 0, words.ctypes.data, 64, res.ctypes.data) == pkg.encoder.E_BAD_ARG
    assert L.jpegenc_encode_coeffs(enc._ctx, None, 8, 8, 0, words.ctypes.data, 64, res.ctypes.data) == pkg.encoder.E_BAD_ARG
    assert L.jpegenc_encode_coeffs(enc._ctx, c.ctypes.data, 8, 8, 0, words.ctypes.data, 0, res.ctypes.data) == pkg.encoder.E_BAD_ARG
