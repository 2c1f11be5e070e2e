"""ctypes binding of the CPU oracle (oracle/libjpegenc_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "libjpegenc_oracle.so")


class RefConfig(C.Structure):
    _fields_ = [("qtab", (C.c_uint16 * 64) * 3),
                ("quirk_rgb_wrap", C.c_uint8), ("quirk_cb_6d64", C.c_uint8),
                ("quirk_ydct_zext", C.c_uint8), ("quant_index_order", C.c_uint8)]


class RefResult(C.Structure):
    _fields_ = [("n_full_words", C.c_uint64), ("total_bits_unstuffed", C.c_uint64),
                ("total_bytes_stuffed", C.c_uint64), ("tail_word", C.c_uint32),
                ("tail_bits", C.c_uint8), ("eof_partial_ready", C.c_uint8), ("pad", C.c_uint8 * 2)]


def build(force=False):
    src = os.path.join(_ORACLE_DIR, "jpegenc_ref.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "libjpegenc_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.ref_default_config.argtypes = [C.POINTER(RefConfig)]
        L.ref_rgb2ycbcr_px.argtypes = [C.POINTER(RefConfig), C.c_uint32, C.POINTER(C.c_uint8)]
        L.ref_dct8x8.argtypes = [C.POINTER(RefConfig), C.c_void_p, C.c_int, C.c_void_p]
        L.ref_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_block_coeffs.argtypes = [C.POINTER(RefConfig), C.c_int, C.c_void_p, C.c_void_p]
        L.ref_huff_block.argtypes = [C.POINTER(RefConfig), C.c_int, C.c_void_p, C.POINTER(C.c_int32),
                                     C.c_void_p, C.c_size_t]
        L.ref_huff_block.restype = C.c_long
        L.ref_max_words.argtypes = [C.c_int, C.c_int]
        L.ref_max_words.restype = C.c_size_t
        L.ref_encode_rgb8.argtypes = [C.POINTER(RefConfig), C.c_void_p, C.c_int, C.c_int, C.c_ssize_t,
                                      C.c_void_p, C.c_size_t, C.POINTER(RefResult)]
        L.ref_encode_rgb8_unstuffed.argtypes = [C.POINTER(RefConfig), C.c_void_p, C.c_int, C.c_int, C.c_ssize_t,
                                                C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.ref_image_coeffs.argtypes = [C.POINTER(RefConfig), C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_void_p]
        L.ref_encode_coeffs.argtypes = [C.POINTER(RefConfig), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.POINTER(RefResult)]
        L.ref_port_open.argtypes = [C.POINTER(RefConfig)]
        L.ref_port_open.restype = C.c_void_p
        L.ref_port_reset.argtypes = [C.c_void_p]
        L.ref_port_push_pixel.argtypes = [C.c_void_p, C.c_uint32]
        L.ref_port_end_of_file.argtypes = [C.c_void_p, C.POINTER(RefResult)]
        L.ref_port_end_of_file.restype = C.c_size_t
        L.ref_port_words.argtypes = [C.c_void_p]
        L.ref_port_words.restype = C.POINTER(C.c_uint32)
        L.ref_port_close.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def default_config(qtabs=None, rgb_wrap=True, cb_6d64=True, ydct_zext=False, quant_index_order=0):
    cfg = RefConfig()
    lib().ref_default_config(C.byref(cfg))
    if qtabs is not None:
        for c in range(3):
            for i in range(64):
                cfg.qtab[c][i] = int(qtabs[c][i])
    cfg.quirk_rgb_wrap = int(rgb_wrap)
    cfg.quirk_cb_6d64 = int(cb_6d64)
    cfg.quirk_ydct_zext = int(ydct_zext)
    cfg.quant_index_order = int(quant_index_order)
    return cfg


def encode(rgb, cfg=None):
    """rgb: (h,w,3) uint8 C-contiguous.  Returns (words uint32 array incl. tail word, RefResult)."""
    cfg = cfg or default_config()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    cap = lib().ref_max_words(w, h)
    words = np.zeros(cap, dtype=np.uint32)
    res = RefResult()
    rc = lib().ref_encode_rgb8(C.byref(cfg), rgb.ctypes.data, w, h, w * 3, words.ctypes.data, cap, C.byref(res))
    if rc:
        raise RuntimeError("ref_encode_rgb8 rc=%d" % rc)
    n = res.n_full_words + (1 if res.tail_bits else 0)
    return words[:n].copy(), res


def image_coeffs(rgb, cfg=None):
    cfg = cfg or default_config()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    out = np.zeros(((h // 8) * (w // 8), 3, 64), dtype=np.int16)
    rc = lib().ref_image_coeffs(C.byref(cfg), rgb.ctypes.data, w, h, w * 3, out.ctypes.data)
    if rc:
        raise RuntimeError("ref_image_coeffs rc=%d" % rc)
    return out


def encode_coeffs(coeffs, cfg=None):
    """coeffs: int16 [n_mcu, 3, 64] in the coders' reading order (what image_coeffs returns).  Returns (words
    incl. tail word, RefResult) of the stream the three Huffman coders make of them after rst."""
    cfg = cfg or default_config()
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
    n_mcu = coeffs.shape[0]
    cap = lib().ref_max_words(8, 8 * n_mcu)
    words = np.zeros(cap, dtype=np.uint32)
    res = RefResult()
    rc = lib().ref_encode_coeffs(C.byref(cfg), coeffs.ctypes.data, n_mcu, words.ctypes.data, cap, C.byref(res))
    if rc:
        raise RuntimeError("ref_encode_coeffs rc=%d" % rc)
    n = res.n_full_words + (1 if res.tail_bits else 0)
    return words[:n].copy(), res


def huff_block(blk_huff_order, comp, pred=0, cfg=None):
    """blk_huff_order: 64 ints, index a*8+b (a vertical).  Returns (bit string, new pred)."""
    cfg = cfg or default_config()
    blk = np.asarray(blk_huff_order, dtype=np.int32).copy()
    bits = np.zeros(4096, dtype=np.uint8)
    p = C.c_int32(pred)
    n = lib().ref_huff_block(C.byref(cfg), comp, blk.ctypes.data, C.byref(p), bits.ctypes.data, bits.size)
    assert n >= 0
    return "".join(str(int(b)) for b in bits[:n]), p.value


def unstuffed_bits(rgb, cfg=None):
    """The image's Huffman bit stream before stuffing, as a uint8 array of 0/1."""
    cfg = cfg or default_config()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    cap = lib().ref_max_words(w, h) * 4
    buf = np.zeros(cap, dtype=np.uint8)
    nbits = C.c_uint64()
    rc = lib().ref_encode_rgb8_unstuffed(C.byref(cfg), rgb.ctypes.data, w, h, w * 3, buf.ctypes.data, cap,
                                         C.byref(nbits))
    if rc:
        raise RuntimeError("ref_encode_rgb8_unstuffed rc=%d" % rc)
    return np.unpackbits(buf)[:nbits.value]


class OracleBand:
    """CPU stand-in for the three segment calls of the C ABI on rows [lo, hi) of `image`, built on the
    oracle: the stream of the first k MCUs is a prefix of the whole image's stream, so a band's bits are
    the difference of two prefix encodings."""

    def __init__(self, image, lo, hi, cfg=None):
        full = unstuffed_bits(image[:hi], cfg)
        skip = len(unstuffed_bits(image[:lo], cfg)) if lo else 0
        self.bits = full[skip:]

    def encode(self):
        return len(self.bits), int(np.packbits(self.bits[-8:])[0])

    def _owned(self):
        s = self.o & 7
        pre = np.unpackbits(np.array([self.prev], dtype=np.uint8))[8 - s:] if s else np.zeros(0, np.uint8)
        allbits = np.concatenate([pre, self.bits])
        nbytes = ((self.o + len(self.bits)) >> 3) - (self.o >> 3)
        return np.packbits(allbits[:8 * nbytes]), allbits[8 * nbytes:]

    def count(self, bit_offset, prev_tail8):
        self.o, self.prev = bit_offset, prev_tail8
        return int((self._owned()[0] == 0xFF).sum())

    def write(self, ff_offset, is_last):
        owned, rest = self._owned()
        out = []
        for b in owned:
            out.append(int(b))
            if b == 0xFF:
                out.append(0)
        pos0 = (self.o >> 3) + ff_offset
        first_word = pos0 >> 2
        result = None
        if is_last:
            stuffed = pos0 + len(out)
            rem = len(rest)
            if rem:
                out.append(int(np.packbits(np.concatenate([rest, np.zeros(8 - rem, np.uint8)]))[0]))
            sbits = stuffed * 8 + rem
            result = dict(n_full_words=sbits >> 5, total_bits_unstuffed=self.o + len(self.bits),
                          total_bytes_stuffed=stuffed, tail_bits=sbits & 31, eof_partial_ready=int(sbits & 31 != 0),
                          status=0)
        lead = pos0 - 4 * first_word
        raw = bytes(lead) + bytes(out)
        raw += bytes((-len(raw)) % 4)
        words = np.frombuffer(raw, dtype=">u4").astype(np.uint32)
        if is_last:
            result["tail_word"] = int(words[result["n_full_words"] - first_word]) if result["tail_bits"] else 0
        return first_word, words, result


class OracleBandResident:
    """OracleBand behind the protocol of a band coder that keeps its figures in tensors (CPU tensors here):
    what `encode_monolithic_resident` drives on the GPUs with the device-resident segment calls."""
    device_resident = True

    def __init__(self, image, lo, hi, cfg=None):
        self.band = OracleBand(image, lo, hi, cfg)

    def encode(self):
        import torch
        return torch.tensor(self.band.encode(), dtype=torch.int64)

    def count(self, bit_offset, prev_tail8):
        import torch
        return torch.tensor([self.band.count(int(bit_offset.item()), int(prev_tail8.item()))], dtype=torch.int64)

    def write(self, ff_offset, is_last):
        import torch
        first_word, words, result = self.band.write(int(ff_offset.item()), is_last)
        rec = np.zeros(32, dtype=np.uint8)
        if result is not None:
            r = np.zeros(1, dtype=[("n_full_words", "<u8"), ("total_bits_unstuffed", "<u8"), ("total_bytes_stuffed", "<u8"),
                                   ("tail_word", "<u4"), ("tail_bits", "u1"), ("eof_partial_ready", "u1"), ("status", "<i2")])
            for k in r.dtype.names:
                r[0][k] = result[k]
            rec = r.view(np.uint8).copy()
        room = np.zeros(len(words) + 5, dtype=np.int32)  # words with room to spare, as the GPU coder returns them
        room[:len(words)] = np.asarray(words, dtype=np.uint32).view(np.int32)
        return (torch.tensor([first_word, len(words)], dtype=torch.int64), torch.from_numpy(rec), torch.from_numpy(room))
