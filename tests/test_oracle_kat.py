"""Hand-derivable known answers for the CPU oracle (and the numpy restatement).

The reference ships no vectors, so these are derived by hand from the RTL arithmetic:
rgb2ycrcb.sv:77-130, cb_dct.sv / y_dct.sv rounding, cr_huff.sv code tables and run logic.
"""
import numpy as np
import pytest

import oracle_lib as O
import ref_numpy as N

ZZ = [r * 8 + c for r, c in N.ZZ]


def block_from_zigzag(z):
    blk = np.zeros(64, dtype=np.int32)
    for k, v in enumerate(z):
        blk[ZZ[k]] = v
    return blk


def numpy_bits(z, comp, pred=0, cb_quirk=True):
    bw = N.BitWriter()
    N.encode_block(z, comp, pred, bw, cb_quirk)
    acc, n = bw.to_int()
    return format(acc, "0%db" % n) if n else ""


def test_flat_grey_64x64():
    # RGB 128: Y=Cb=Cr=128 exactly, every coefficient 0; per block DC size 0 '00' + EOB '1010'.
    rgb = np.full((64, 64, 3), 128, dtype=np.uint8)
    words, res = O.encode(rgb)
    assert res.total_bits_unstuffed == 64 * 18
    assert res.n_full_words == 36 and res.tail_bits == 0 and res.eof_partial_ready == 0
    pat = [0x28A28A28, 0xA28A28A2, 0x8A28A28A]
    assert list(words) == [pat[i % 3] for i in range(36)]
    ref = N.encode_image(rgb)
    assert ref["words"] == list(words)


def test_colour_conversion_known_values():
    cfg = O.default_config()
    out = (O.C.c_uint8 * 3)()

    def conv(r, g, b, c=cfg):
        O.lib().ref_rgb2ycbcr_px(O.C.byref(c), r | (g << 8) | (b << 16), out)
        return tuple(out)

    assert conv(128, 128, 128) == (128, 128, 128)
    assert conv(0, 0, 0) == (0, 128, 128)
    assert conv(255, 255, 255) == (255, 128, 128)
    # Q1: the 8-bit rounder wraps 255+1 to 0 -- only pure blue (Cb) and pure red (Cr) reach it.
    assert conv(0, 0, 255)[1] == 0
    assert conv(255, 0, 0)[2] == 0
    off = O.default_config(rgb_wrap=False)
    assert conv(0, 0, 255, off)[1] == 255
    assert conv(255, 0, 0, off)[2] == 255


def test_colour_conversion_exhaustive_vs_numpy():
    # all 2^24 inputs through numpy in slabs; the oracle on a strided sample plus the extremes
    r, g = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    wraps_cb = wraps_cr = 0
    for b in (0, 1, 127, 128, 254, 255):
        rgb = np.stack([r, g, np.full_like(r, b)], axis=-1)
        ycc = N.rgb_to_ycbcr(rgb)
        noq = N.rgb_to_ycbcr(rgb, wrap_quirk=False)
        wraps_cb += int(np.sum(ycc[..., 1] != noq[..., 1]))
        wraps_cr += int(np.sum(ycc[..., 2] != noq[..., 2]))
        assert np.array_equal(ycc[..., 0], noq[..., 0])  # Y can never wrap
    assert wraps_cb == 1 and wraps_cr == 1
    cfg = O.default_config()
    out = (O.C.c_uint8 * 3)()
    rng = np.random.default_rng(5)
    for px in rng.integers(0, 1 << 24, 5000):
        px = int(px)
        O.lib().ref_rgb2ycbcr_px(O.C.byref(cfg), px, out)
        rgb = np.array([[[px & 255, (px >> 8) & 255, px >> 16]]], dtype=np.uint8)
        assert tuple(out) == tuple(N.rgb_to_ycbcr(rgb)[0, 0])


def test_flat_block_dc_values():
    cfg = O.default_config()
    F = np.zeros(64, dtype=np.int32)

    def dc(val, luma):
        x = np.full(64, val, dtype=np.uint8)
        O.lib().ref_dct8x8(O.C.byref(cfg), x.ctypes.data, int(luma), F.ctypes.data)
        assert not F[1:].any()
        return int(F[0])

    # flat black luma: A=-5932032, B=rhu(A,12)=-1448, Z=8*(-1448*5793), F=rhu(Z,16)=-1024
    assert dc(0, True) == -1024
    assert dc(0, False) == -1024
    assert dc(128, True) == 0 and dc(128, False) == 0
    # the two intermediate precisions differ on flat white
    assert dc(255, True) == 1016
    assert dc(255, False) == 1015


def test_flat_black_first_block_bits():
    # DC diff -1024: size 11 -> code 11111111110, amplitude (-1025)&0x7FF = 01111111111, then EOB.
    rgb = np.zeros((8, 16, 3), dtype=np.uint8)
    coeffs = O.image_coeffs(rgb)
    assert coeffs[0, 0, 0] == -1024 and coeffs[0, 1, 0] == 0 and coeffs[0, 2, 0] == 0
    bits, pred = O.huff_block(block_from_zigzag([-1024] + [0] * 63), 0, 0)
    assert bits == "11111111110" + "01111111111" + "1010"
    assert pred == -1024
    # second luma block: difference 0
    bits2, _ = O.huff_block(block_from_zigzag([-1024] + [0] * 63), 0, pred)
    assert bits2 == "00" + "1010"


def test_dct_matches_float_dct():
    scipy_fft = pytest.importorskip("scipy.fft")
    rng = np.random.default_rng(11)
    cfg = O.default_config()
    F = np.zeros(64, dtype=np.int32)
    for luma, tol in ((True, 0.9), (False, 1.8)):
        for _ in range(20):
            x = rng.integers(0, 256, 64, dtype=np.uint8)
            O.lib().ref_dct8x8(O.C.byref(cfg), x.ctypes.data, int(luma), F.ctypes.data)
            want = scipy_fft.dctn(x.reshape(8, 8).astype(np.float64) - 128, norm="ortho")  # [vertical][horizontal]
            got = F.reshape(8, 8).T  # F[k][j] with k horizontal -> transpose to [vertical][horizontal]
            assert np.max(np.abs(got - want)) <= tol


def test_quantiser_default_is_identity_and_reciprocal_rule():
    F = np.arange(-1024, 1024, 32, dtype=np.int32)[:64].copy()
    Q = np.zeros(64, dtype=np.int32)
    ones = np.ones(64, dtype=np.uint16)
    O.lib().ref_quantize(F.ctypes.data, ones.ctypes.data, Q.ctypes.data)
    assert np.array_equal(F, Q)
    q = np.full(64, 3, dtype=np.uint16)  # QM = 4096 // 3 = 1365, not a true division
    O.lib().ref_quantize(F.ctypes.data, q.ctypes.data, Q.ctypes.data)
    want = (F.astype(np.int64) * 1365 + 2048) >> 12
    assert np.array_equal(Q, want)
    q[:] = 0
    O.lib().ref_quantize(F.ctypes.data, q.ctypes.data, Q.ctypes.data)
    assert not Q.any()


# ---- entropy coder corner cases (coefficient level, bypassing the DCT) --------------------------
ZRL = "11111111001"
EOB = "1010"


def ac(sym):
    code, length = N.AC_TAB[sym]
    return format(code, "0%db" % length)


@pytest.mark.parametrize("comp", [0, 1, 2])
def test_plain_run_and_eob(comp):
    z = [0] * 64
    z[1] = 1
    z[5] = -2
    bits, _ = O.huff_block(block_from_zigzag(z), comp)
    want = "00" + ac(0x01) + "1" + ac(0x32) + "01" + EOB
    assert bits == want == numpy_bits(z, comp)


@pytest.mark.parametrize("comp", [0, 1, 2])
def test_zrl_interior_run(comp):
    z = [0] * 64
    z[20] = 3  # 19 zeros before it: one ZRL (16th zero at k=16, bc=18 <32, later coefficients exist)
    bits, _ = O.huff_block(block_from_zigzag(z), comp)
    want = "00" + ZRL + ac(0x32) + "11" + EOB
    assert bits == want == numpy_bits(z, comp)


@pytest.mark.parametrize("comp", [0, 1, 2])
def test_q2_dropped_zrl_at_k30(comp):
    # 16th zero lands on k=30, z[31] != 0, z[32..63] all zero: the coarse end-of-block test is
    # true there and the ZRL is dropped; the coefficient is coded with run mod 16 = 0.
    z = [0] * 64
    z[14] = 1
    z[31] = 1
    bits, _ = O.huff_block(block_from_zigzag(z), comp)
    want = "00" + ac(0xD1) + "1" + ac(0x01) + "1" + EOB
    assert bits == want == numpy_bits(z, comp)
    # with anything non-zero in 32..63 the ZRL is emitted as usual
    z[40] = 1
    bits, _ = O.huff_block(block_from_zigzag(z), comp)
    want = "00" + ac(0xD1) + "1" + ZRL + ac(0x01) + "1" + ac(0x81) + "1" + EOB
    assert bits == want == numpy_bits(z, comp)


def test_q2_q3_at_k46():
    z = [0] * 64
    z[30] = 1
    z[47] = 1  # 16th zero after z[30] lands on k=46, z[48..63] zero
    for comp in (0, 1, 2):
        bits, _ = O.huff_block(block_from_zigzag(z), comp)
        want = "00" + ZRL + ac(0xD1) + "1" + ac(0x01) + "1" + EOB
        assert bits == want == numpy_bits(z, comp)
    # with a later non-zero coefficient Y/Cr emit the ZRL; Cb (6'd64 literal) still drops it
    z[50] = 1
    for comp in (0, 2):
        bits, _ = O.huff_block(block_from_zigzag(z), comp)
        want = "00" + ZRL + ac(0xD1) + "1" + ZRL + ac(0x01) + "1" + ac(0x21) + "1" + EOB
        assert bits == want == numpy_bits(z, comp)
    bits, _ = O.huff_block(block_from_zigzag(z), 1)
    want = "00" + ZRL + ac(0xD1) + "1" + ac(0x01) + "1" + ac(0x21) + "1" + EOB
    assert bits == want == numpy_bits(z, 1)
    # switch Q3 off: Cb behaves like Y/Cr
    off = O.default_config(cb_6d64=False)
    bits, _ = O.huff_block(block_from_zigzag(z), 1, cfg=off)
    want = "00" + ZRL + ac(0xD1) + "1" + ZRL + ac(0x01) + "1" + ac(0x21) + "1" + EOB
    assert bits == want == numpy_bits(z, 1, cb_quirk=False)


@pytest.mark.parametrize("comp", [0, 1, 2])
def test_last_coefficient_nonzero_has_no_eob(comp):
    z = [0] * 64
    z[63] = 5
    z[60] = -1
    bits, _ = O.huff_block(block_from_zigzag(z), comp)
    # zeros 1..59 = 59 zeros -> ZRLs at k=16,32,48 for Y/Cr (later non-zero exists); Cb drops k=48
    zrls = ZRL * (2 if comp == 1 else 3)
    want = "00" + zrls + ac(0xB1) + "0" + ac(0x23) + "101"
    assert bits == want == numpy_bits(z, comp)


@pytest.mark.parametrize("comp", [0, 1, 2])
def test_trailing_zero_run_never_emits_zrl(comp):
    for last in (1, 13, 14, 15, 29, 30, 31, 45, 46, 47):
        z = [0] * 64
        z[last] = 1
        bits, _ = O.huff_block(block_from_zigzag(z), comp)
        # interior run before z[last]: one ZRL per 16 zeros; the trailing run emits none
        want = "00" + ZRL * ((last - 1) // 16) + ac((((last - 1) % 16) << 4) | 1) + "1" + EOB
        assert bits == want == numpy_bits(z, comp)


def test_q5_minus_1024_size_zero_symbol():
    z = [0] * 64
    z[1] = -1024  # size10 = 0 while v != 0: symbol (run<<4)|0 with no amplitude; run 0 -> EOB code
    bits, _ = O.huff_block(block_from_zigzag(z), 0)
    assert bits == "00" + EOB + EOB == numpy_bits(z, 0)
    z = [0] * 64
    z[3] = -1024  # run 2 -> map entry (2<<4) -> table entry 0 = code '00'
    bits, _ = O.huff_block(block_from_zigzag(z), 0)
    assert bits == "00" + "00" + EOB == numpy_bits(z, 0)


def test_dc_difference_wraps_in_12_bits_and_sizes():
    for prev, cur in ((0, 1023), (1023, -1024), (-1024, 1023), (5, 5), (0, -1), (0, 1)):
        bits, pred = O.huff_block(block_from_zigzag([cur] + [0] * 63), 2, prev)
        assert pred == cur
        assert bits == numpy_bits([cur] + [0] * 63, 2, prev)
    bits, _ = O.huff_block(block_from_zigzag([1023] + [0] * 63), 0, -1024)  # d = 2047, size 11
    assert bits == "11111111110" + "11111111111" + EOB


def test_stuffing_and_tail():
    # 5 bits of 1s then FF byte: build via pack_stream directly and via a crafted block stream
    r = N.pack_stream(0xFFFF, 16)
    assert r["words"] == [0xFF00FF00] and r["tail_bits"] == 0 and r["n_full_words"] == 1
    r = N.pack_stream(0xFF, 8)
    assert r["words"] == [0xFF000000] and r["tail_bits"] == 16 and r["n_full_words"] == 0
    r = N.pack_stream(0x7F, 7)  # incomplete byte is never stuffed
    assert r["words"] == [0xFE000000] and r["tail_bits"] == 7
    r = N.pack_stream((0xFF << 3) | 0b101, 11)
    assert r["words"] == [0xFF00A000] and r["tail_bits"] == 19


def test_port_shim_matches_raster_api():
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (16, 24, 3), dtype=np.uint8)
    words, res = O.encode(rgb)
    L = O.lib()
    cfg = O.default_config()
    p = L.ref_port_open(O.C.byref(cfg))
    L.ref_port_reset(p)
    for by in range(2):
        for bx in range(3):
            for r in range(8):
                for c in range(8):
                    px = rgb[by * 8 + r, bx * 8 + c]
                    L.ref_port_push_pixel(p, int(px[0]) | (int(px[1]) << 8) | (int(px[2]) << 16))
    r2 = O.RefResult()
    n = L.ref_port_end_of_file(p, O.C.byref(r2))
    got = np.ctypeslib.as_array(L.ref_port_words(p), shape=(n,)).copy()
    L.ref_port_close(p)
    assert n == len(words) and np.array_equal(got, words)
    assert (r2.n_full_words, r2.tail_bits, r2.tail_word) == (res.n_full_words, res.tail_bits, res.tail_word)
