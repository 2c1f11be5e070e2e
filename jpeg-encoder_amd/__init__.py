"""jpeg-encoder_amd: the jpeg_top data path (RGB -> stuffed JPEG scan words) on MI355X.

The directory name is not a Python identifier; import it with
    importlib.import_module("jpeg-encoder_amd")
(bench.py, __graft_entry__.py and tests/ do exactly that through `load_pkg()` below when this file
is executed as part of the package)."""
from . import distributed, jfif, synth  # noqa: F401
from .encoder import (ABI_SYMBOLS, DEVICE_PTRS, E_AGAIN, E_BAD_ARG, E_HIP, E_LIMIT, E_NOMEM, E_STATE, E_TOO_SMALL, E_UNSUPPORTED, HOST_PTRS,  # noqa: F401
                      LIB_PATH, NOALLOC_MAX_CHUNKS, OK, PAD_EDGES, RESULT_DTYPE, SEG_CONTINUES, Config, JpegEncError, JpegTop, Result, load_library, max_words,
                      raster_to_port_order, words_to_bytes)
