"""JFIF/JPEG container around a jpeg_top scan stream (host side, no compute).

jpeg_top emits only entropy-coded scan data (rtl/0_jpeg_top/jpeg_top.sv:18-28 has no marker logic), so
a viewer needs the headers added: SOI, APP0, DQT, SOF0 (8 bit, 3 components, 1x1 sampling), DHT, SOS,
the scan bytes, EOI.  The Huffman tables are the ones hard-coded in rtl/4_huffman/*_huff.sv -- T.81 K.4
for DC and K.5 for AC, shared by all three components -- and the DQT entries are the configured
quantisation tables in zig-zag order (all ones by default).
"""
import struct

import numpy as np

DC_BITS = [0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0]
DC_VALS = list(range(12))
AC_BITS = [0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7D]
AC_VALS = list(bytes.fromhex(
    "01020300041105122131410613516107227114328191a1082342b1c11552d1f0243362728209"
    "0a161718191a25262728292a3435363738393a434445464748494a535455565758595a636465"
    "666768696a737475767778797a838485868788898a92939495969798999aa2a3a4a5a6a7a8a9"
    "aab2b3b4b5b6b7b8b9bac2c3c4c5c6c7c8c9cad2d3d4d5d6d7d8d9dae1e2e3e4e5e6e7e8e9ea"
    "f1f2f3f4f5f6f7f8f9fa"))

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14,
          21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53,
          60, 61, 54, 47, 55, 62, 63]


def _segment(marker, payload):
    return struct.pack(">BBH", 0xFF, marker, len(payload) + 2) + payload


def scan_bytes(words, n_full_words, tail_bits):
    """The stuffed scan as bytes; the unused bits of the last byte are filled with ones (T.81 F.1.2.3)."""
    nbits = int(n_full_words) * 32 + int(tail_bits)
    raw = bytearray(np.asarray(words, dtype=np.uint32).astype(">u4").tobytes()[:(nbits + 7) // 8])
    rem = nbits % 8
    if rem:
        raw[-1] |= (1 << (8 - rem)) - 1
        if raw[-1] == 0xFF:
            raw.append(0x00)
    return bytes(raw)


def wrap_jfif(words, n_full_words, tail_bits, width, height, qtabs=None, quant_index_order=0):
    """Returns a complete baseline JPEG file.  qtabs: three 64-entry tables in the configuration's
    order (quant_index_order 0: entry [i*8+j] scales horizontal frequency i, vertical frequency j; 1:
    entry [a*8+b] scales vertical a, horizontal b), default all ones."""
    out = bytearray(b"\xFF\xD8")
    out += _segment(0xE0, b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    for c in range(3):
        q = [1] * 64 if qtabs is None else [int(v) for v in qtabs[c]]
        # zig-zag position k is (row a = vertical, col b = horizontal) = ZIGZAG[k]; config index b*8+a
        if quant_index_order:
            zz = [min(255, max(1, q[ZIGZAG[k]])) for k in range(64)]
        else:
            zz = [min(255, max(1, q[(ZIGZAG[k] % 8) * 8 + ZIGZAG[k] // 8])) for k in range(64)]
        out += _segment(0xDB, bytes([c]) + bytes(zz))
    sof = struct.pack(">BHHB", 8, height, width, 3)
    for c in range(3):
        sof += bytes([c + 1, 0x11, c])
    out += _segment(0xC0, sof)
    out += _segment(0xC4, bytes([0x00]) + bytes(DC_BITS) + bytes(DC_VALS))
    out += _segment(0xC4, bytes([0x10]) + bytes(AC_BITS) + bytes(AC_VALS))
    out += _segment(0xDA, bytes([3, 1, 0x00, 2, 0x00, 3, 0x00, 0, 63, 0]))
    out += scan_bytes(words, n_full_words, tail_bits)
    out += b"\xFF\xD9"
    return bytes(out)
