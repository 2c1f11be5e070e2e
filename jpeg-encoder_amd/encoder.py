"""ctypes binding of libjpegenc.so -- the jpeg_top data path on MI355X.

The reference's only interface is the port list of rtl/0_jpeg_top/jpeg_top.sv:18-28; the method names
below follow those ports (reset = rst, push_pixel = enable + data_in, end_of_file =
end_of_file_signal, words = JPEG_bitstream under data_ready, tail_bits =
end_of_file_bitstream_count, eof_partial_ready = eof_data_partial_ready).  Everything computes in
the HIP library; if it cannot be loaded or no gfx950 device is usable, calls raise -- there is no
CPU path in this package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjpegenc.so")

OK, E_BAD_ARG, E_TOO_SMALL, E_HIP, E_UNSUPPORTED, E_NOMEM, E_STATE, E_LIMIT, E_AGAIN = 0, -1, -2, -3, -4, -5, -6, -7, -8
NOALLOC_MAX_CHUNKS = 2048
HOST_PTRS, DEVICE_PTRS = 0, 1
PAD_EDGES = 4
SEG_CONTINUES = 2
N_KERNELS = 5

# every symbol include/jpegenc.h declares
ABI_SYMBOLS = [
    "jpegenc_default_config", "jpegenc_create", "jpegenc_destroy", "jpegenc_set_stream", "jpegenc_synchronize",
    "jpegenc_max_words", "jpegenc_encode_rgb8", "jpegenc_encode_batch", "jpegenc_encode_batch_noalloc",
    "jpegenc_encode_stripes", "jpegenc_image_coeffs", "jpegenc_encode_coeffs", "jpegenc_reset",
    "jpegenc_push_pixel", "jpegenc_push_pixels", "jpegenc_end_of_file", "jpegenc_read_words",
    "jpegenc_words_to_bytes", "jpegenc_set_profiling", "jpegenc_kernel_times", "jpegenc_kernel_name",
    "jpegenc_set_workspace_limit", "jpegenc_set_slot_mode", "jpegenc_workspace_bytes", "jpegenc_set_region_mode", "jpegenc_region_stats", "jpegenc_set_pipeline", "jpegenc_reserve", "jpegenc_segment_encode", "jpegenc_segment_count",
    "jpegenc_segment_write", "jpegenc_segment_encode_dev", "jpegenc_segment_count_dev", "jpegenc_segment_write_dev", "jpegenc_strerror", "jpegenc_last_error", "jpegenc_version",
]


class Config(C.Structure):
    _fields_ = [("qtab", (C.c_uint16 * 64) * 3),
                ("quirk_rgb_wrap", C.c_uint8), ("quirk_cb_6d64", C.c_uint8),
                ("quirk_ydct_zext", C.c_uint8), ("quant_index_order", C.c_uint8),
                ("device", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("n_full_words", C.c_uint64), ("total_bits_unstuffed", C.c_uint64),
                ("total_bytes_stuffed", C.c_uint64), ("tail_word", C.c_uint32),
                ("tail_bits", C.c_uint8), ("eof_partial_ready", C.c_uint8), ("status", C.c_int16)]


RESULT_DTYPE = np.dtype([("n_full_words", "<u8"), ("total_bits_unstuffed", "<u8"), ("total_bytes_stuffed", "<u8"),
                         ("tail_word", "<u4"), ("tail_bits", "u1"), ("eof_partial_ready", "u1"), ("status", "<i2")])
assert RESULT_DTYPE.itemsize == C.sizeof(Result) == 32


class JpegEncError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("jpegenc error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library(path=None):
    """Loads libjpegenc.so (raises OSError if it has not been built)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime under the same sonames as /opt/rocm.  Two
    # runtimes in one process do not both see the GPU, so when torch is installed let it load its copy
    # first: libjpegenc.so's NEEDED libamdhip64.so.7 then binds to that one and device pointers,
    # streams and events are shared with torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path or LIB_PATH)
    vp, sz, i32, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
    L.jpegenc_default_config.argtypes = [C.POINTER(Config)]
    L.jpegenc_create.argtypes = [C.POINTER(vp), C.POINTER(Config)]
    L.jpegenc_destroy.argtypes = [vp]
    L.jpegenc_set_stream.argtypes = [vp, vp]
    L.jpegenc_synchronize.argtypes = [vp]
    L.jpegenc_max_words.argtypes = [i32, i32]
    L.jpegenc_max_words.restype = sz
    L.jpegenc_encode_rgb8.argtypes = [vp, vp, i32, i32, C.c_ssize_t, u32, vp, sz, vp]
    L.jpegenc_encode_batch.argtypes = [vp, vp, i32, i32, i32, C.c_ssize_t, C.c_ssize_t, u32, vp, sz, vp]
    L.jpegenc_encode_batch_noalloc.argtypes = [vp, vp, i32, i32, i32, C.c_ssize_t, C.c_ssize_t, vp, sz, vp]
    L.jpegenc_encode_stripes.argtypes = [vp, vp, i32, i32, C.c_ssize_t, i32, u32, vp, sz, vp]
    L.jpegenc_reset.argtypes = [vp]
    L.jpegenc_push_pixel.argtypes = [vp, u32]
    L.jpegenc_push_pixels.argtypes = [vp, vp, sz]
    L.jpegenc_end_of_file.argtypes = [vp, C.POINTER(Result)]
    L.jpegenc_read_words.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.jpegenc_words_to_bytes.argtypes = [vp, C.POINTER(Result), vp, sz, C.POINTER(sz)]
    L.jpegenc_set_profiling.argtypes = [vp, i32]
    L.jpegenc_kernel_times.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_double), i32]
    L.jpegenc_kernel_name.argtypes = [i32]
    L.jpegenc_kernel_name.restype = C.c_char_p
    L.jpegenc_set_workspace_limit.argtypes = [vp, sz]
    L.jpegenc_set_slot_mode.argtypes = [vp, i32]
    L.jpegenc_workspace_bytes.argtypes = [vp]
    L.jpegenc_workspace_bytes.restype = sz
    L.jpegenc_reserve.argtypes = [vp, i32, i32, i32]
    L.jpegenc_set_region_mode.argtypes = [vp, i32]
    L.jpegenc_set_pipeline.argtypes = [vp, i32]
    L.jpegenc_region_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.jpegenc_image_coeffs.argtypes = [vp, vp, i32, i32, C.c_ssize_t, u32, vp]
    L.jpegenc_encode_coeffs.argtypes = [vp, vp, i32, i32, u32, vp, sz, vp]
    u64 = C.c_uint64
    L.jpegenc_segment_encode.argtypes = [vp, vp, i32, i32, C.c_ssize_t, u32, C.POINTER(u64), C.POINTER(u32)]
    L.jpegenc_segment_count.argtypes = [vp, u64, u32, C.POINTER(u64)]
    L.jpegenc_segment_write.argtypes = [vp, u64, i32, u32, vp, sz, C.POINTER(u64), C.POINTER(u64), C.POINTER(Result)]
    L.jpegenc_segment_encode_dev.argtypes = [vp, vp, i32, i32, C.c_ssize_t, u32, vp]
    L.jpegenc_segment_count_dev.argtypes = [vp, vp, vp, vp]
    L.jpegenc_segment_write_dev.argtypes = [vp, vp, i32, vp, sz, vp, vp]
    L.jpegenc_strerror.argtypes = [i32]
    L.jpegenc_strerror.restype = C.c_char_p
    L.jpegenc_last_error.argtypes = [vp]
    L.jpegenc_last_error.restype = C.c_char_p
    L.jpegenc_version.restype = C.c_char_p
    if path is None:
        _lib = L
    return L


def max_words(w, h):
    return int(load_library().jpegenc_max_words(w, h))


class JpegTop:
    """One instance of the jpeg_top path bound to a GPU (one context = one device + one stream)."""

    def __init__(self, qtabs=None, rgb_wrap=True, cb_6d64=True, device=0, stream=None, quant_index_order=0):
        self.lib = load_library()
        cfg = Config()
        self.lib.jpegenc_default_config(C.byref(cfg))
        if qtabs is not None:
            for c in range(3):
                for i in range(64):
                    cfg.qtab[c][i] = int(qtabs[c][i])
        cfg.quirk_rgb_wrap = int(rgb_wrap)
        cfg.quirk_cb_6d64 = int(cb_6d64)
        cfg.quant_index_order = int(quant_index_order)
        cfg.device = int(device)
        self.device = int(device)
        self._ctx = C.c_void_p()
        rc = self.lib.jpegenc_create(C.byref(self._ctx), C.byref(cfg))
        if rc != OK:
            self._ctx = None
            raise JpegEncError(rc, self.lib.jpegenc_strerror(rc).decode())
        if stream is not None:
            self.set_stream(stream)

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc, allow=()):
        if rc != OK and rc not in allow:
            detail = self.lib.jpegenc_last_error(self._ctx).decode()
            raise JpegEncError(rc, self.lib.jpegenc_strerror(rc).decode() + (" (" + detail + ")" if detail else ""))
        return rc

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.jpegenc_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    HIP_STREAM_LEGACY = 1  # hipStreamLegacy, the handle that names the device's default stream

    def set_stream(self, hip_stream):
        """hip_stream: integer handle (e.g. torch.cuda.current_stream().cuda_stream), or None for the
        context's own stream.  torch reports the device's default stream as handle 0, which the C ABI reads
        as "own stream" (a non-blocking one, not ordered with the default stream): it is passed on as
        hipStreamLegacy, so that work torch has queued on its default stream is ordered before the encoder's."""
        if hip_stream is not None and int(hip_stream) == 0:
            hip_stream = self.HIP_STREAM_LEGACY
        self._check(self.lib.jpegenc_set_stream(self._ctx, C.c_void_p(hip_stream)))

    def synchronize(self):
        self._check(self.lib.jpegenc_synchronize(self._ctx))

    def set_workspace_limit(self, nbytes):
        self._check(self.lib.jpegenc_set_workspace_limit(self._ctx, int(nbytes)))

    SLOTS_WORST_CASE, SLOTS_ADAPTIVE = 0, 1

    def set_slot_mode(self, mode):
        """SLOTS_WORST_CASE (default) or SLOTS_ADAPTIVE: scratch per group of 64 blocks sized from the longest group
        seen so far; images of a call that outgrows it come back with status E_AGAIN (device-pointer calls)."""
        self._check(self.lib.jpegenc_set_slot_mode(self._ctx, int(mode)))

    def workspace_bytes(self):
        return int(self.lib.jpegenc_workspace_bytes(self._ctx))

    REGIONS_AUTO, REGIONS_SMALL, REGIONS_LARGE, REGIONS_SHORT = 0, 1, 2, 3

    def set_region_mode(self, mode):
        """LDS staging regions of the encode kernel: REGIONS_AUTO (default), REGIONS_SMALL, REGIONS_LARGE,
        REGIONS_SHORT; speed only."""
        self._check(self.lib.jpegenc_set_region_mode(self._ctx, int(mode)))

    PIPELINE_AUTO, PIPELINE_SPLIT, PIPELINE_SINGLE_MCU = 0, 1, 2

    def set_pipeline(self, mode):
        """PIPELINE_AUTO (default): small calls run as one kernel launch; PIPELINE_SPLIT: always separate kernels."""
        self._check(self.lib.jpegenc_set_pipeline(self._ctx, int(mode)))

    def region_stats(self):
        """(groups with a block too long for a small region, ... for a large region, groups, groups with a block too
        long for a short region) of the last launch that collected them, and the size REGIONS_AUTO stands at."""
        c = (C.c_uint64 * 5)()
        self._check(self.lib.jpegenc_region_stats(self._ctx, c))
        return int(c[0]), int(c[1]), int(c[2]), int(c[3]), int(c[4])

    def reserve(self, n_images, w, h):
        """Grows the workspace now, so that device-pointer calls of up to this size only enqueue work."""
        self._check(self.lib.jpegenc_reserve(self._ctx, int(n_images), int(w), int(h)))

    def set_profiling(self, on):
        self._check(self.lib.jpegenc_set_profiling(self._ctx, int(bool(on))))

    def kernel_times(self, reset=True):
        n = (C.c_uint64 * N_KERNELS)()
        ms = (C.c_double * N_KERNELS)()
        self._check(self.lib.jpegenc_kernel_times(self._ctx, n, ms, int(reset)))
        return {self.lib.jpegenc_kernel_name(i).decode(): (int(n[i]), float(ms[i])) for i in range(N_KERNELS)}

    # -- raster API (host memory) -----------------------------------------------------------
    def encode(self, rgb, cap_words=None, pad_edges=False):
        """rgb: (h, w, 3) uint8 numpy array.  Returns (words incl. tail word, result record)."""
        words, res = self.encode_batch(rgb[None], cap_words, pad_edges=pad_edges)
        return words[0], res[0]

    def encode_batch(self, rgb, cap_words=None, allow_too_small=False, pad_edges=False):
        """rgb: (n, h, w, 3) uint8 numpy array.  Returns (list of word arrays, results array).
        pad_edges (an extension the RTL does not have): h and w need not be multiples of 8, the last row
        and column are repeated up to the next multiples and that padded image is what gets coded."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        n, h, w, _ = rgb.shape
        cap = int(cap_words or max_words(w, h))
        words = np.zeros((n, cap), dtype=np.uint32)
        res = np.zeros(n, dtype=RESULT_DTYPE)
        rc = self.lib.jpegenc_encode_batch(self._ctx, rgb.ctypes.data, n, w, h, w * 3, h * w * 3,
                                           HOST_PTRS | (PAD_EDGES if pad_edges else 0),
                                           words.ctypes.data, cap, res.ctypes.data)
        self._check(rc, allow=(E_TOO_SMALL,) if allow_too_small else ())
        out = []
        for i in range(n):
            k = min(cap, int(res[i]["n_full_words"]) + (1 if res[i]["tail_bits"] else 0))
            out.append(words[i, :k].copy())
        return out, res

    def encode_stripes(self, rgb, n_stripes, cap_words=None):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, _ = rgb.shape
        cap = int(cap_words or max_words(w, h // n_stripes))
        words = np.zeros((n_stripes, cap), dtype=np.uint32)
        res = np.zeros(n_stripes, dtype=RESULT_DTYPE)
        self._check(self.lib.jpegenc_encode_stripes(self._ctx, rgb.ctypes.data, w, h, w * 3, n_stripes, HOST_PTRS,
                                                    words.ctypes.data, cap, res.ctypes.data))
        return [words[i, :int(res[i]["n_full_words"]) + (1 if res[i]["tail_bits"] else 0)].copy()
                for i in range(n_stripes)], res

    def image_coeffs(self, rgb):
        """The coefficients as they enter the Huffman coders: int16 array [blocks, 3, 64] (block row-major,
        component Y/Cb/Cr, position in the coder's zig-zag reading order).  rgb: uint8 [h, w, 3] on the host."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, _ = rgb.shape
        out = np.zeros(((h // 8) * (w // 8), 3, 64), dtype=np.int16)
        self._check(self.lib.jpegenc_image_coeffs(self._ctx, rgb.ctypes.data, w, h, w * 3, HOST_PTRS, out.ctypes.data))
        return out

    def encode_coeffs(self, coeffs, w, h, cap_words=None):
        """The stream the three Huffman coders, the interleave and the stuffing make of `coeffs` (int16
        [blocks, 3, 64] as image_coeffs returns them, any 11-bit values) for a w x h image after rst.
        Returns (words incl. tail word, result record)."""
        coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
        assert coeffs.shape == ((h // 8) * (w // 8), 3, 64)
        cap = int(cap_words or max_words(w, h))
        words = np.zeros(cap, dtype=np.uint32)
        res = np.zeros(1, dtype=RESULT_DTYPE)
        self._check(self.lib.jpegenc_encode_coeffs(self._ctx, coeffs.ctypes.data, w, h, HOST_PTRS, words.ctypes.data,
                                                   cap, res.ctypes.data))
        k = min(cap, int(res[0]["n_full_words"]) + (1 if res[0]["tail_bits"] else 0))
        return words[:k].copy(), res[0]

    # -- raster API (device memory, asynchronous on the context's stream) -------------------
    def encode_batch_device(self, rgb_ptr, n, w, h, stride, image_stride, words_ptr, cap_words, results_ptr):
        """All three pointers are device addresses (ints); only enqueues work."""
        self._check(self.lib.jpegenc_encode_batch(self._ctx, C.c_void_p(rgb_ptr), n, w, h, stride, image_stride,
                                                  DEVICE_PTRS, C.c_void_p(words_ptr), cap_words,
                                                  C.c_void_p(results_ptr)))

    def encode_batch_noalloc(self, rgb_ptr, n, w, h, stride, image_stride, words_ptr, cap_words, results_ptr):
        """jpegenc_encode_batch_noalloc: as encode_batch_device, with the library's static workspace (at most
        NOALLOC_MAX_CHUNKS chunks of 64 blocks per call, JpegEncError with code E_LIMIT beyond)."""
        self._check(self.lib.jpegenc_encode_batch_noalloc(self._ctx, C.c_void_p(rgb_ptr), n, w, h, stride, image_stride,
                                                          C.c_void_p(words_ptr), cap_words, C.c_void_p(results_ptr)))

    def encode_stripes_device(self, rgb_ptr, w, h, stride, n_stripes, words_ptr, cap_words_per_stripe, results_ptr):
        """jpegenc_encode_stripes on device addresses (ints); only enqueues work."""
        self._check(self.lib.jpegenc_encode_stripes(self._ctx, C.c_void_p(rgb_ptr), w, h, stride, n_stripes,
                                                    DEVICE_PTRS, C.c_void_p(words_ptr), cap_words_per_stripe,
                                                    C.c_void_p(results_ptr)))

    # -- one stream from consecutive bands of an image -----------------------------------------
    def segment_encode(self, image, row0, rows):
        """Codes rows [row0, row0 + rows) of the (h, w, 3) uint8 host array `image` as a band of the
        image's single stream.  Returns (bits, tail8)."""
        assert image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3 and image.flags.c_contiguous
        h, w, _ = image.shape
        assert 0 <= row0 and row0 + rows <= h and row0 % 8 == 0
        bits, tail = C.c_uint64(), C.c_uint32()
        ptr = image.ctypes.data + row0 * w * 3
        self._check(self.lib.jpegenc_segment_encode(self._ctx, ptr, w, rows, w * 3,
                                                    HOST_PTRS | (SEG_CONTINUES if row0 else 0),
                                                    C.byref(bits), C.byref(tail)))
        return int(bits.value), int(tail.value)

    def segment_encode_device(self, rgb_ptr, w, rows, stride, continues):
        """Same with a device pointer to the band's first row (the 8 rows above it must be readable when
        `continues`)."""
        bits, tail = C.c_uint64(), C.c_uint32()
        self._check(self.lib.jpegenc_segment_encode(self._ctx, C.c_void_p(rgb_ptr), w, rows, stride,
                                                    DEVICE_PTRS | (SEG_CONTINUES if continues else 0),
                                                    C.byref(bits), C.byref(tail)))
        return int(bits.value), int(tail.value)

    def segment_count(self, bit_offset, prev_tail8):
        ff = C.c_uint64()
        self._check(self.lib.jpegenc_segment_count(self._ctx, int(bit_offset), int(prev_tail8), C.byref(ff)))
        return int(ff.value)

    def segment_write(self, ff_offset, is_last, cap_words):
        """Returns (first_word, words, result or None): words[i] is word first_word + i of the whole
        stream; bytes of its first and last word that belong to the neighbouring bands are zero."""
        words = np.zeros(cap_words, dtype=np.uint32)
        res = Result()
        fw, nw = C.c_uint64(), C.c_uint64()
        self._check(self.lib.jpegenc_segment_write(self._ctx, int(ff_offset), int(bool(is_last)), HOST_PTRS,
                                                   words.ctypes.data, cap_words, C.byref(fw), C.byref(nw),
                                                   C.byref(res)))
        rec = None
        if is_last:
            rec = np.zeros(1, dtype=RESULT_DTYPE)
            C.memmove(rec.ctypes.data, C.byref(res), 32)
            rec = rec[0]
        return int(fw.value), words[:int(nw.value)], rec

    def segment_write_device(self, ff_offset, is_last, words_ptr, cap_words):
        """Same with the words left in device memory at `words_ptr` (room for cap_words).  Returns
        (first_word, n_words, result or None)."""
        res = Result()
        fw, nw = C.c_uint64(), C.c_uint64()
        self._check(self.lib.jpegenc_segment_write(self._ctx, int(ff_offset), int(bool(is_last)), DEVICE_PTRS,
                                                   C.c_void_p(words_ptr), cap_words, C.byref(fw), C.byref(nw),
                                                   C.byref(res)))
        rec = None
        if is_last:
            rec = np.zeros(1, dtype=RESULT_DTYPE)
            C.memmove(rec.ctypes.data, C.byref(res), 32)
            rec = rec[0]
        return int(fw.value), int(nw.value), rec

    # the same three steps with every figure left in and read from device memory: only enqueue work
    def segment_encode_dev(self, rgb_ptr, w, rows, stride, continues, bits_tail8_ptr):
        """bits_tail8_ptr: device address of two uint64 (bit count, last 8 bits)."""
        self._check(self.lib.jpegenc_segment_encode_dev(self._ctx, C.c_void_p(rgb_ptr), w, rows, stride,
                                                        DEVICE_PTRS | (SEG_CONTINUES if continues else 0),
                                                        C.c_void_p(bits_tail8_ptr)))

    def segment_count_dev(self, bit_offset_ptr, prev_tail8_ptr, ff_bytes_ptr):
        """Device addresses of one uint64 each: two read, one written."""
        self._check(self.lib.jpegenc_segment_count_dev(self._ctx, C.c_void_p(bit_offset_ptr), C.c_void_p(prev_tail8_ptr),
                                                       C.c_void_p(ff_bytes_ptr)))

    def segment_write_dev(self, ff_offset_ptr, is_last, words_ptr, cap_words, span_ptr, result_ptr):
        """span_ptr: device address of two uint64 (first_word, n_words); result_ptr: of a 32-byte result record."""
        self._check(self.lib.jpegenc_segment_write_dev(self._ctx, C.c_void_p(ff_offset_ptr), int(bool(is_last)),
                                                       C.c_void_p(words_ptr), cap_words, C.c_void_p(span_ptr),
                                                       C.c_void_p(result_ptr)))

    # -- port-level protocol -----------------------------------------------------------------
    def reset(self):
        self._check(self.lib.jpegenc_reset(self._ctx))

    def push_pixel(self, data_in24):
        self._check(self.lib.jpegenc_push_pixel(self._ctx, int(data_in24) & 0xFFFFFF))

    def push_pixels(self, data_in24):
        a = np.ascontiguousarray(data_in24, dtype=np.uint32)
        self._check(self.lib.jpegenc_push_pixels(self._ctx, a.ctypes.data, a.size))

    def end_of_file(self):
        """Returns (words, Result) -- the data_ready words followed by the partial word, if any."""
        res = Result()
        self._check(self.lib.jpegenc_end_of_file(self._ctx, C.byref(res)))
        n = C.c_size_t()
        self.lib.jpegenc_read_words(self._ctx, None, 0, C.byref(n))
        words = np.zeros(max(1, n.value), dtype=np.uint32)
        self._check(self.lib.jpegenc_read_words(self._ctx, words.ctypes.data, words.size, C.byref(n)))
        return words[:n.value].copy(), res


def words_to_bytes(words, n_full_words, tail_bits):
    """Big-endian byte view of the word stream, truncated to the valid bits (rounded up to bytes)."""
    nbits = int(n_full_words) * 32 + int(tail_bits)
    return np.asarray(words, dtype=np.uint32).astype(">u4").tobytes()[:(nbits + 7) // 8]


def raster_to_port_order(rgb):
    """(h, w, 3) uint8 -> data_in[23:0] values in the order jpeg_top expects them: block by block,
    left to right, top to bottom, 64 pixels per block row-major, {B,G,R} with R in bits 7:0."""
    h, w, _ = rgb.shape
    blocks = rgb.reshape(h // 8, 8, w // 8, 8, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 3).astype(np.uint32)
    return blocks[:, 0] | (blocks[:, 1] << 8) | (blocks[:, 2] << 16)
