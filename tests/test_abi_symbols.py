"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/jpegenc.h declares,
and its host-only helpers behave.  No compute call is made (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import numpy as np

from conftest import ROOT, load_pkg


def header_symbols():
    text = open(os.path.join(ROOT, "include", "jpegenc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jpegenc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    pkg = load_pkg()
    lib = C.CDLL(pkg.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libjpegenc.so does not export %s" % s
    assert sorted(pkg.ABI_SYMBOLS) == syms


def test_struct_layouts_match_header():
    pkg = load_pkg()
    assert C.sizeof(pkg.Result) == 32
    assert C.sizeof(pkg.Config) == 3 * 64 * 2 + 4 + 4
    cfg = pkg.Config()
    assert pkg.load_library().jpegenc_default_config(C.byref(cfg)) == pkg.OK
    assert all(cfg.qtab[c][i] == 1 for c in range(3) for i in range(64))
    assert (cfg.quirk_rgb_wrap, cfg.quirk_cb_6d64, cfg.quirk_ydct_zext, cfg.device) == (1, 1, 0, 0)


def test_host_only_helpers():
    pkg = load_pkg()
    lib = pkg.load_library()
    # worst case: 4980 bits per MCU, doubled by stuffing, plus the tail word
    assert pkg.max_words(8, 8) == (4980 * 2 + 31) // 32 + 1
    assert pkg.max_words(1920, 1080) == (32400 * 4980 * 2 + 31) // 32 + 1
    assert lib.jpegenc_strerror(pkg.E_TOO_SMALL) == b"output buffer too small"
    res = pkg.Result()
    res.n_full_words, res.tail_bits = 1, 12
    words = np.array([0x11223344, 0xABC00000], dtype=np.uint32)
    out = np.zeros(8, dtype=np.uint8)
    n = C.c_size_t()
    assert lib.jpegenc_words_to_bytes(words.ctypes.data, C.byref(res), out.ctypes.data, 8, C.byref(n)) == pkg.OK
    assert n.value == 6 and bytes(out[:6]) == bytes([0x11, 0x22, 0x33, 0x44, 0xAB, 0xC0])
    assert pkg.words_to_bytes(words, 1, 12) == bytes([0x11, 0x22, 0x33, 0x44, 0xAB, 0xC0])
    assert lib.jpegenc_words_to_bytes(words.ctypes.data, C.byref(res), out.ctypes.data, 3, C.byref(n)) == pkg.E_TOO_SMALL


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    pkg = load_pkg()
    lib = pkg.load_library()
    assert lib.jpegenc_default_config(None) == pkg.E_BAD_ARG
    assert lib.jpegenc_create(None, None) == pkg.E_BAD_ARG
    assert lib.jpegenc_destroy(None) == pkg.E_BAD_ARG
    cfg = pkg.Config()
    lib.jpegenc_default_config(C.byref(cfg))
    cfg.quirk_ydct_zext = 1
    ctx = C.c_void_p()
    assert lib.jpegenc_create(C.byref(ctx), C.byref(cfg)) == pkg.E_UNSUPPORTED
    # every call that takes a context refuses a null one before it touches HIP
    null = C.c_void_p()
    assert lib.jpegenc_set_stream(null, None) == pkg.E_BAD_ARG
    assert lib.jpegenc_synchronize(null) == pkg.E_BAD_ARG
    assert lib.jpegenc_reserve(null, 1, 8, 8) == pkg.E_BAD_ARG
    assert lib.jpegenc_set_pipeline(null, 0) == pkg.E_BAD_ARG
    assert lib.jpegenc_set_region_mode(null, 0) == pkg.E_BAD_ARG
    assert lib.jpegenc_set_workspace_limit(null, 1 << 20) == pkg.E_BAD_ARG
    assert lib.jpegenc_encode_batch_noalloc(null, None, 1, 8, 8, 24, 192, None, 16, None) == pkg.E_BAD_ARG
    assert lib.jpegenc_encode_batch(null, None, 1, 8, 8, 24, 192, 0, None, 16, None) == pkg.E_BAD_ARG
    assert lib.jpegenc_segment_count(null, 0, 0, None) == pkg.E_BAD_ARG
    assert lib.jpegenc_image_coeffs(null, None, 8, 8, 24, 0, None) == pkg.E_BAD_ARG
    assert lib.jpegenc_region_stats(null, None) == pkg.E_BAD_ARG
    assert lib.jpegenc_strerror(pkg.E_LIMIT) == b"call exceeds the static workspace of jpegenc_encode_batch_noalloc"
    assert lib.jpegenc_strerror(pkg.E_NOMEM) == b"out of memory"


def test_header_constants_match_the_python_mirror():
    """The limits and codes the Python class repeats are the header's."""
    pkg = load_pkg()
    text = open(os.path.join(ROOT, "include", "jpegenc.h")).read()
    defs = dict(re.findall(r"#define\s+(JPEGENC_[A-Z0-9_]+)\s+\(?(-?\d+)u?\)?", text))
    assert int(defs["JPEGENC_NOALLOC_MAX_CHUNKS"]) == pkg.NOALLOC_MAX_CHUNKS
    assert int(defs["JPEGENC_N_KERNELS"]) == 5
    assert (int(defs["JPEGENC_E_LIMIT"]), int(defs["JPEGENC_E_NOMEM"]), int(defs["JPEGENC_E_STATE"])) == \
           (pkg.E_LIMIT, pkg.E_NOMEM, pkg.E_STATE)
    assert (int(defs["JPEGENC_PIPELINE_AUTO"]), int(defs["JPEGENC_PIPELINE_SPLIT"]), int(defs["JPEGENC_PIPELINE_SINGLE_MCU"])) == \
           (pkg.JpegTop.PIPELINE_AUTO, pkg.JpegTop.PIPELINE_SPLIT, pkg.JpegTop.PIPELINE_SINGLE_MCU)
    assert (int(defs["JPEGENC_REGIONS_AUTO"]), int(defs["JPEGENC_REGIONS_SMALL"]), int(defs["JPEGENC_REGIONS_LARGE"]),
            int(defs["JPEGENC_REGIONS_SHORT"])) == \
           (pkg.JpegTop.REGIONS_AUTO, pkg.JpegTop.REGIONS_SMALL, pkg.JpegTop.REGIONS_LARGE, pkg.JpegTop.REGIONS_SHORT)
    assert (int(defs["JPEGENC_DEVICE_PTRS"]), int(defs["JPEGENC_PAD_EDGES"]), int(defs["JPEGENC_SEG_CONTINUES"])) == \
           (pkg.DEVICE_PTRS, pkg.PAD_EDGES, pkg.SEG_CONTINUES)


def test_port_order_helper():
    pkg = load_pkg()
    rgb = np.arange(16 * 16 * 3, dtype=np.uint32).reshape(16, 16, 3).astype(np.uint8)
    d = pkg.raster_to_port_order(rgb)
    assert d.size == 256
    # first block, second row starts at raster pixel (1, 0)
    px = rgb[1, 0]
    assert d[8] == int(px[0]) | (int(px[1]) << 8) | (int(px[2]) << 16)
    # second block starts at raster pixel (0, 8)
    px = rgb[0, 8]
    assert d[64] == int(px[0]) | (int(px[1]) << 8) | (int(px[2]) << 16)
