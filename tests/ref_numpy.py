"""Second, independently written restatement of the jpeg_top data path (numpy + Python loops).

TEST INFRASTRUCTURE ONLY.  It exists to cross-check oracle/jpegenc_ref.cpp: it was written from
the formula sheet of the path (fixed-point colour conversion, two-pass integer DCT with
intermediate rounding, reciprocal quantiser, JPEG-style entropy coder with the RTL's quirks,
0xFF00 stuffing, left-aligned tail), not from the C++ text, and shares no code or tables with
it or with the HIP path.  Reference citations (paths relative to the reference tree):
colour rtl/1_rbg2ycrcb/rgb2ycrcb.sv:29-39,65-137; DCT rtl/2_dct/cb_dct.sv, rtl/2_dct/y_dct.sv;
quantiser rtl/3_quantization/y_quantizer.sv:37-46,58,74,106; transpose rtl/d_q_h/yd_q_h.sv:120-135;
entropy rtl/4_huffman/cr_huff.sv (+ y_huff.sv:184-193, cb_huff.sv:230-243 for the end-of-block
thresholds); interleave rtl/5_fifo/fifo_out.sv:195-197,222-243; stuffing rtl/5_fifo/ff_checker.sv.
"""
import math

import numpy as np

# ---------------------------------------------------------------- constants
T = np.array([[round(math.cos((2 * c + 1) * k * math.pi / 16) * (math.sqrt(0.125) if k == 0 else 0.5) * 16384)
               for c in range(8)] for k in range(8)], dtype=np.int64)
# The RTL's constants are these roundings except that it uses 5793 for both 0.35355*2^14 entries
# (rows 0 and 4), 7568/3135 and 8035/6811/4551/1598; assert we reproduced them.
assert T[0, 0] == 5793 and T[4, 1] == -5793 and list(T[1, :4]) == [8035, 6811, 4551, 1598]
assert list(T[2, :2]) == [7568, 3135]

LEVEL_SHIFT = 128 * 8 * 5793


def zigzag_order():
    """T.81 Figure A.6 traversal generated algorithmically: list of (row, col)."""
    out = []
    for s in range(15):
        rng = range(0, s + 1)
        diag = [(i, s - i) for i in rng if i < 8 and s - i < 8]
        out.extend(diag if s % 2 else diag[::-1])
    return out


ZZ = zigzag_order()
assert ZZ[:6] == [(0, 0), (0, 1), (1, 0), (2, 0), (1, 1), (0, 2)] and ZZ[-1] == (7, 7)

DC_BITS = [0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0]
AC_BITS = [0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7D]
AC_VALS = bytes.fromhex(
    "01020300041105122131410613516107227114328191a1082342b1c11552d1f0243362728209"
    "0a161718191a25262728292a3435363738393a434445464748494a535455565758595a636465"
    "666768696a737475767778797a838485868788898a92939495969798999aa2a3a4a5a6a7a8a9"
    "aab2b3b4b5b6b7b8b9bac2c3c4c5c6c7c8c9cad2d3d4d5d6d7d8d9dae1e2e3e4e5e6e7e8e9ea"
    "f1f2f3f4f5f6f7f8f9fa")
assert len(AC_VALS) == 162


def canonical(bits, vals):
    table = {}
    code = 0
    it = iter(vals)
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            table[next(it)] = (code, length)
            code += 1
        code <<= 1
    return table


DC_TAB = canonical(DC_BITS, range(12))
AC_TAB = canonical(AC_BITS, AC_VALS)
FIRST_AC = AC_TAB[AC_VALS[0]]  # entry 0 of the RTL's code-ordered table
for _run in range(1, 15):
    AC_TAB[_run << 4] = FIRST_AC  # the RTL maps (run 1..14, size 0) to table entry 0


# ---------------------------------------------------------------- arithmetic helpers
def wrap(x, n):
    x = np.asarray(x, dtype=np.int64) & ((1 << n) - 1)
    return np.where(x >> (n - 1), x - (1 << n), x)


def rhu(x, n):
    x = np.asarray(x, dtype=np.int64)
    return (x >> n) + ((x >> (n - 1)) & 1)


def rgb_to_ycbcr(rgb, wrap_quirk=True):
    """rgb: (...,3) uint8 -> (...,3) uint8 planes Y, Cb, Cr."""
    r = rgb[..., 0].astype(np.int64)
    g = rgb[..., 1].astype(np.int64)
    b = rgb[..., 2].astype(np.int64)
    sums = np.stack([4899 * r + 9617 * g + 1868 * b,
                     2097152 - 2764 * r - 5428 * g + 8192 * b,
                     2097152 + 8192 * r - 6860 * g - 1332 * b], axis=-1)
    v = rhu(sums, 14)
    v = (v & 0xFF) if wrap_quirk else np.minimum(v, 255)
    return v.astype(np.uint8)


def dct_blocks(x, luma):
    """x: (n,8,8) uint8 [row][col] -> (n,8,8) int64 F[k][j] (k horizontal, j vertical frequency)."""
    x = x.astype(np.int64)
    a = np.einsum("nrc,kc->nrk", x, T)
    a[:, :, 0] -= LEVEL_SHIFT
    a = wrap(a, 25)
    b = wrap(rhu(a, 12), 13) if luma else wrap(rhu(a, 14), 11)
    z = np.einsum("nrk,jr->nkj", b, T)
    if luma:
        return wrap(rhu(wrap(z, 32), 16), 11)
    return wrap(rhu(wrap(z, 25), 14), 11)


def quantize(f, qtab):
    """f: (n,8,8) F[k][j]; qtab: 64 entries indexed k*8+j."""
    q = np.asarray(qtab, dtype=np.int64).reshape(8, 8)
    qm = np.where(q == 0, 0, 4096 // np.maximum(q, 1))
    prod = wrap(wrap(f, 11) * qm, 23)
    return wrap((prod + 2048) >> 12, 11)


def image_coeffs(rgb, qtabs=None, wrap_quirk=True):
    """rgb (h,w,3) uint8 -> (nMCU, 3, 64) int64 coefficients in zig-zag order."""
    h, w, _ = rgb.shape
    assert h % 8 == 0 and w % 8 == 0
    ycc = rgb_to_ycbcr(rgb, wrap_quirk)
    blocks = ycc.reshape(h // 8, 8, w // 8, 8, 3).transpose(0, 2, 4, 1, 3).reshape(-1, 3, 8, 8)
    out = np.zeros((blocks.shape[0], 3, 64), dtype=np.int64)
    for c in range(3):
        f = dct_blocks(blocks[:, c], luma=(c == 0))
        if qtabs is not None:
            f = quantize(f, qtabs[c])
        hu = f.transpose(0, 2, 1)  # wrapper transpose: huff[a][b] = Q[b][a]
        for k, (row, col) in enumerate(ZZ):
            out[:, c, k] = hu[:, row, col]
    return out


# ---------------------------------------------------------------- entropy coder
def bit_length_low(v, nbits):
    """Size rule of the RTL: look only at the low `nbits` bits of v (v>=0) or of v-1 (v<0)."""
    mask = (1 << nbits) - 1
    if v >= 0:
        return (v & mask).bit_length()
    return (~(v - 1) & mask).bit_length()


class BitWriter:
    def __init__(self):
        self.chunks = []
        self.n = 0

    def put(self, code, length):
        if length:
            self.chunks.append((code & ((1 << length) - 1), length))
            self.n += length

    def to_int(self):
        acc = 0
        for code, length in self.chunks:
            acc = (acc << length) | code
        return acc, self.n


def encode_block(z, comp, pred, bw, cb_quirk=True):
    """z: 64 ints in zig-zag order.  Returns the new DC predictor."""
    z = [int(v) for v in z]
    d = int(wrap(z[0] - pred, 12))
    s = bit_length_low(d, 11)
    bw.put(*DC_TAB[s])
    bw.put((d if d >= 0 else d - 1), s)
    zero8 = [all(v == 0 for v in z[8 * g:8 * g + 8]) for g in range(8)]

    def eob(k):
        bc = k + 2
        if bc < 32:
            return all(zero8[2:8])
        if bc < 48:
            return all(zero8[4:8])
        if comp == 1 and cb_quirk:
            return True
        if bc <= 64:
            return all(zero8[6:8])
        return True

    run = 0
    for k in range(1, 63):
        v = z[k]
        if v == 0:
            if run == 15 and not eob(k):
                bw.put(*AC_TAB[0xF0])
            run = (run + 1) & 15
        else:
            s = bit_length_low(v, 10)
            bw.put(*AC_TAB[(run << 4) | s])
            bw.put((v if v >= 0 else v - 1), s)
            run = 0
    v = z[63]
    if v == 0:
        bw.put(*AC_TAB[0x00])
    else:
        s = bit_length_low(v, 10)
        bw.put(*AC_TAB[(run << 4) | s])
        bw.put((v if v >= 0 else v - 1), s)
    return z[0]


def encode_image(rgb, qtabs=None, wrap_quirk=True, cb_quirk=True):
    """Returns dict(words=list[int], n_full_words, tail_bits, tail_word, total_bits, stuffed_bytes)."""
    coeffs = image_coeffs(rgb, qtabs, wrap_quirk)
    bw = BitWriter()
    pred = [0, 0, 0]
    for b in range(coeffs.shape[0]):
        for c in range(3):
            pred[c] = encode_block(coeffs[b, c], c, pred[c], bw, cb_quirk)
    acc, nbits = bw.to_int()
    return pack_stream(acc, nbits)


def pack_stream(acc, nbits):
    full = nbits // 8
    rem = nbits % 8
    raw = (acc << ((8 - rem) % 8)).to_bytes(full + (1 if rem else 0), "big") if nbits else b""
    st = bytearray()
    for byte in raw[:full]:
        st.append(byte)
        if byte == 0xFF:
            st.append(0)
    stuffed = len(st)
    if rem:
        st.append(raw[full])
    sbits = stuffed * 8 + rem
    while len(st) % 4:
        st.append(0)
    words = [int.from_bytes(st[i:i + 4], "big") for i in range(0, len(st), 4)]
    n_full = sbits // 32
    tail_bits = sbits % 32
    words = words[:n_full + (1 if tail_bits else 0)]
    return dict(words=words, n_full_words=n_full, tail_bits=tail_bits,
                tail_word=(words[n_full] if tail_bits else 0), total_bits=nbits, stuffed_bytes=stuffed)
