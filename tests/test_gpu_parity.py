"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

The bar is exact equality of every 32-bit word, of the tail word/count and of the sizes.  The oracle
itself is pinned only by the RTL text, an independent numpy restatement and hand-derived known
answers (the reference ships no vectors): see oracle/jpegenc_ref.h.
"""
import numpy as np
import pytest

import oracle_lib as O
from conftest import load_pkg, natural_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["auto", "single_mcu", "split"])
def enc(request):
    """Small calls run as one kernel launch by default (the smallest with a lane per component block, the others
    with a lane per block of all three) and as separate kernels on request: all three forms take every test that
    uses this encoder."""
    pkg = load_pkg()
    e = pkg.JpegTop()
    if request.param == "split":
        e.set_pipeline(e.PIPELINE_SPLIT)
    if request.param == "single_mcu":
        e.set_pipeline(e.PIPELINE_SINGLE_MCU)
    yield e
    e.close()


def assert_same(words, res, rgb, cfg=None):
    want, r = O.encode(rgb, cfg)
    assert int(res["total_bits_unstuffed"]) == r.total_bits_unstuffed
    assert int(res["total_bytes_stuffed"]) == r.total_bytes_stuffed
    assert int(res["n_full_words"]) == r.n_full_words
    assert int(res["tail_bits"]) == r.tail_bits
    assert int(res["eof_partial_ready"]) == r.eof_partial_ready
    assert int(res["tail_word"]) == r.tail_word
    assert int(res["status"]) == 0
    assert len(words) == len(want)
    if not np.array_equal(words, want):
        bad = int(np.nonzero(words != want)[0][0])
        raise AssertionError("first differing word %d of %d: got %08x want %08x" % (bad, len(want), words[bad], want[bad]))


def config1_images():
    y, x = np.mgrid[0:64, 0:64]
    return [np.full((64, 64, 3), 128, np.uint8),
            np.stack([(4 * x) & 255, (4 * y) & 255, (2 * (x + y)) & 255], axis=-1).astype(np.uint8),
            np.random.default_rng(1).integers(0, 256, (64, 64, 3), dtype=np.uint8)]


def test_config1_64x64_raster(enc):
    for rgb in config1_images():
        words, res = enc.encode(rgb)
        assert_same(words, res, rgb)
    words, res = enc.encode(config1_images()[0])
    pat = [0x28A28A28, 0xA28A28A2, 0x8A28A28A]
    assert list(words) == [pat[i % 3] for i in range(36)] and int(res["tail_bits"]) == 0


def test_config1_64x64_through_the_port_protocol(enc):
    pkg = load_pkg()
    for rgb in config1_images():
        enc.reset()
        enc.push_pixels(pkg.raster_to_port_order(rgb))
        words, res = enc.end_of_file()
        want, r = O.encode(rgb)
        assert np.array_equal(words, want)
        assert (res.n_full_words, res.tail_bits, res.tail_word, res.eof_partial_ready) == \
               (r.n_full_words, r.tail_bits, r.tail_word, r.eof_partial_ready)
    # pixel-at-a-time, and protocol errors
    rgb = config1_images()[2][:8, :16]
    enc.reset()
    for d in pkg.raster_to_port_order(rgb):
        enc.push_pixel(int(d))
    words, res = enc.end_of_file()
    assert np.array_equal(words, O.encode(rgb)[0])
    with pytest.raises(pkg.JpegEncError):
        enc.push_pixel(0)  # rst required after end_of_file
    enc.reset()
    words, res = enc.end_of_file()  # nothing pushed: nothing comes out
    assert len(words) == 0 and res.n_full_words == 0 and res.tail_bits == 0


@pytest.mark.parametrize("hw", [(8, 8), (8, 16), (16, 8), (24, 40), (8, 512), (8, 520), (72, 72), (128, 8), (200, 328),
                                (24, 512), (40, 520), (16, 1016), (24, 504)])
def test_ragged_sizes_random(enc, hw):
    # covers: a single MCU, partial chunks, chunks wrapping over block rows, one-MCU-wide images, and block
    # rows of exactly 64, just over 64, 127 and 63 MCUs (the kernel derives a lane's MCU coordinates from the
    # chunk's first MCU with one step to the next block row when a row has at least 64 MCUs)
    h, w = hw
    rgb = np.random.default_rng(h * 1000 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    words, res = enc.encode(rgb)
    assert_same(words, res, rgb)


@pytest.mark.parametrize("hw", [(8, 8 * 40000), (8 * 40000, 8), (16, 8 * 20001), (8 * 9999, 24)])
def test_extreme_aspect_ratios(enc, hw):
    """One block row of 40 000 MCUs (625 chunks, none wraps), a column of 40 000 block rows of one MCU (every lane of
    every chunk in a row of its own), and two shapes whose chunks straddle rows at every phase: the MCU coordinates of
    a lane come from the chunk's first MCU by at most one step to the next row only where a row has 64 MCUs or more."""
    h, w = hw
    rng = np.random.default_rng(h + 7 * w)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # every other 8x8 tile smooth: short and long MCU strings side by side within a chunk
    ty, tx = np.mgrid[0:h, 0:w]
    smooth = ((ty // 8 + tx // 8) & 1).astype(bool)
    rgb[smooth] = (rgb[smooth] >> 5) + 100
    words, res = enc.encode(rgb)
    assert_same(words, res, rgb)


def test_tiny_last_chunk_shares_its_word(enc):
    """65 MCUs = one full chunk and one of a single flat MCU (18 bits): the image's last word, the one
    reported as tail_word, then mostly belongs to the first chunk (found by tools/fuzz_gpu.py)."""
    for seed in (14, 18, 19, 29, 32, 35, 0, 1):  # the first six put the last chunk's bytes into the shared word
        rgb = np.random.default_rng(700 + seed).integers(0, 256, (8, 520, 3), dtype=np.uint8)
        rgb[:, 504:] = rgb[0, 504]  # the last two MCUs flat and equal: zero DC differences in the last one
        words, res = enc.encode(rgb)
        assert_same(words, res, rgb)
    big = np.kron(np.random.default_rng(3).integers(0, 2, (33, 33, 3)) * 255, np.ones((8, 8, 1))).astype(np.uint8)
    words, res = enc.encode_batch(np.stack([big, big[::-1].copy()]))
    for i, img in enumerate((big, big[::-1])):
        assert_same(words[i], res[i], np.ascontiguousarray(img))


def test_more_images_than_one_grid_dimension_holds(enc):
    """The encode kernel's grid carries the image index in y and z (65535 each): 65 540 one-MCU images."""
    rng = np.random.default_rng(65540)
    imgs = rng.integers(0, 256, (65540, 8, 8, 3), dtype=np.uint8)
    words, res = enc.encode_batch(imgs)
    assert len(words) == len(imgs) and all(int(r["status"]) == 0 for r in res)
    for i in [0, 1, 65533, 65534, 65535, 65536, 65539] + rng.integers(0, 65540, 24).tolist():
        assert_same(words[i], res[i], imgs[i])


def test_natural_and_noise_1080p(enc):
    rgb = natural_like(1080, 1920, 2)
    words, res = enc.encode(rgb)
    assert_same(words, res, rgb)
    rgb = np.random.default_rng(3).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    words, res = enc.encode(rgb)
    assert_same(words, res, rgb)
    assert int(res["total_bytes_stuffed"]) > int(res["total_bits_unstuffed"]) // 8  # 0xFF bytes were stuffed


def test_extreme_patterns_take_the_long_string_path(enc):
    # 0/255 noise produces the longest codes: some MCUs exceed the 63-word fast-path capacity
    rng = np.random.default_rng(7)
    rgb = rng.choice(np.array([0, 255], dtype=np.uint8), size=(64, 256, 3))
    words, res = enc.encode(rgb)
    assert_same(words, res, rgb)
    assert int(res["total_bits_unstuffed"]) > 64 * 256 // 64 * 2016  # average above the fast-path capacity
    # mixed: only a few long MCUs inside otherwise smooth content
    rgb2 = natural_like(64, 256, 5)
    rgb2[8:24, 40:72] = rgb[8:24, 40:72]
    words, res = enc.encode(rgb2)
    assert_same(words, res, rgb2)
    # saturated colours hit the 8-bit wrap of the colour rounder
    for ch in (0, 2):
        rgb3 = np.zeros((16, 64, 3), np.uint8)
        rgb3[..., ch] = 255
        words, res = enc.encode(rgb3)
        assert_same(words, res, rgb3)
    y, x = np.mgrid[0:32, 0:64]
    chk = (((x + y) & 1) * 255).astype(np.uint8)[..., None].repeat(3, -1)
    words, res = enc.encode(chk)
    assert_same(words, res, chk)


def test_quirk_switches_and_quant_tables():
    pkg = load_pkg()
    rng = np.random.default_rng(9)
    rgb = natural_like(64, 96, 3)
    rgb[:8, :8, 0], rgb[:8, :8, 1], rgb[:8, :8, 2] = 0, 0, 255  # pure blue block
    noise = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    qt = [rng.integers(1, 64, 64), rng.integers(1, 100, 64), np.full(64, 16)]
    qt[0][5] = 0
    sparse = [np.full(64, 255) for _ in range(3)]
    for c in range(3):
        sparse[c][rng.integers(0, 64, 6)] = rng.integers(1, 8, 6)
    cases = [dict(), dict(rgb_wrap=False), dict(cb_6d64=False), dict(qtabs=qt), dict(qtabs=sparse),
             dict(qtabs=sparse, cb_6d64=False), dict(qtabs=qt, quant_index_order=1)]
    streams = {}
    for k, kw in enumerate(cases):
        with pkg.JpegTop(**kw) as e:
            cfg = O.default_config(qtabs=kw.get("qtabs"), rgb_wrap=kw.get("rgb_wrap", True),
                                   cb_6d64=kw.get("cb_6d64", True), quant_index_order=kw.get("quant_index_order", 0))
            for name, img in (("rgb", rgb), ("noise", noise)):
                words, res = e.encode(img)
                assert_same(words, res, img, cfg)
                streams[k, name] = words
    # every switch must matter on at least one of the images, or the comparisons above prove nothing about it
    def differ(a, b):
        return any(not np.array_equal(streams[a, n], streams[b, n]) for n in ("rgb", "noise"))
    assert differ(0, 1), "rgb_wrap changed nothing"
    assert differ(0, 2), "cb_6d64 changed nothing at the default tables"
    assert differ(4, 5), "cb_6d64 changed nothing at the sparse tables"
    assert differ(3, 6), "quant_index_order changed nothing"


def test_coefficients_before_the_entropy_coders():
    """The stage in front of the Huffman coders on its own (jpegenc_image_coeffs against the oracle's
    ref_image_coeffs): colour conversion, both DCT passes, quantisers and transpose, value for value -- natural
    content, noise, the colours that wrap or clamp, flat extremes; default and non-default tables in both index
    orders, both settings of the colour-rounder switch.  A stream mismatch can then be placed before or after
    this point."""
    pkg = load_pkg()
    rng = np.random.default_rng(77)
    nat = natural_like(72, 136, 8)
    nat[:8, :8] = (0, 0, 255)     # pure blue: Cb reaches 256
    nat[8:16, :8] = (255, 0, 0)   # pure red: Cr reaches 256
    nat[16:24, :8] = 255
    nat[24:32, :8] = 0
    noise = rng.integers(0, 256, (40, 264, 3), dtype=np.uint8)
    checker = np.zeros((16, 16, 3), np.uint8)
    checker[::2, 1::2] = 255
    checker[1::2, ::2] = 255
    qt = [rng.integers(1, 64, 64), rng.integers(1, 100, 64), np.full(64, 16)]
    qt[1][9] = 0
    cases = [dict(), dict(rgb_wrap=False), dict(qtabs=qt), dict(qtabs=qt, quant_index_order=1),
             dict(qtabs=qt, rgb_wrap=False)]
    seen = {}
    for k, kw in enumerate(cases):
        cfg = O.default_config(qtabs=kw.get("qtabs"), rgb_wrap=kw.get("rgb_wrap", True),
                               quant_index_order=kw.get("quant_index_order", 0))
        with pkg.JpegTop(**kw) as e:
            for name, img in (("nat", nat), ("noise", noise), ("checker", checker)):
                got = e.image_coeffs(img)
                want = O.image_coeffs(img, cfg)
                assert got.shape == want.shape
                if not np.array_equal(got, want):
                    blk, comp, idx = (int(v[0]) for v in np.nonzero(got != want))
                    raise AssertionError("%s case %d: block %d component %d index %d: got %d want %d" % (
                        name, k, blk, comp, idx, got[blk, comp, idx], want[blk, comp, idx]))
                seen[k, name] = got
    assert not np.array_equal(seen[0, "nat"], seen[1, "nat"]), "rgb_wrap changed no coefficient"
    assert not np.array_equal(seen[2, "noise"], seen[3, "noise"]), "quant_index_order changed no coefficient"
    # device pointers: the same values, enqueued on the context's stream
    torch = pytest.importorskip("torch")
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        d_img = torch.from_numpy(noise).cuda()
        d_out = torch.zeros((5 * 33, 3, 64), dtype=torch.int16, device="cuda")
        rc = e.lib.jpegenc_image_coeffs(e._ctx, d_img.data_ptr(), 264, 40, 264 * 3, pkg.DEVICE_PTRS,
                                        d_out.data_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), seen[0, "noise"])
        # sizes that are not multiples of 8 are refused
        assert e.lib.jpegenc_image_coeffs(e._ctx, d_img.data_ptr(), 260, 40, 264 * 3, pkg.DEVICE_PTRS,
                                          d_out.data_ptr()) == pkg.E_BAD_ARG


def test_batch_and_stripes(enc):
    rng = np.random.default_rng(21)
    imgs = np.stack([natural_like(72, 136, 30 + i) for i in range(5)] +
                    [rng.integers(0, 256, (72, 136, 3), dtype=np.uint8)])
    words, res = enc.encode_batch(imgs)
    for i in range(len(imgs)):
        assert_same(words[i], res[i], imgs[i])
    # small workspace: forces several sub-batches
    enc.set_workspace_limit(64 * 156 * 4 * 3 * 2)
    words2, res2 = enc.encode_batch(imgs)
    enc.set_workspace_limit(4 << 30)
    for a, b in zip(words, words2):
        assert np.array_equal(a, b)
    assert res.tobytes() == res2.tobytes()
    # stripes: each stripe is the encoding of that sub-image after rst
    big = natural_like(128, 264, 77)
    words, res = enc.encode_stripes(big, 4)
    for s in range(4):
        assert_same(words[s], res[s], big[32 * s:32 * (s + 1)])


def merged_bands(pkg, img, cuts, device_ptrs=False):
    """Runs the three segment calls for every band (one context per band, as one GPU per band would) and
    ORs the bands' words."""
    bands = [pkg.JpegTop() for _ in cuts[:-1]]
    try:
        h, w, _ = img.shape
        if device_ptrs:
            import torch
            d_img = torch.from_numpy(img).to("cuda:0")
            torch.cuda.synchronize()
            info = [e.segment_encode_device(d_img.data_ptr() + a * w * 3, w, b - a, w * 3, a > 0)
                    for e, a, b in zip(bands, cuts[:-1], cuts[1:])]
        else:
            info = [e.segment_encode(img, a, b - a) for e, a, b in zip(bands, cuts[:-1], cuts[1:])]
        offs = np.concatenate([[0], np.cumsum([i[0] for i in info])])
        ffs = [e.segment_count(int(offs[i]), info[i - 1][1] if i else 0) for i, e in enumerate(bands)]
        ffo = np.concatenate([[0], np.cumsum(ffs)])
        out = np.zeros(pkg.max_words(w, h), dtype=np.uint32)
        res = None
        for i, (e, a, b) in enumerate(zip(bands, cuts[:-1], cuts[1:])):
            fw, words, res = e.segment_write(int(ffo[i]), i == len(bands) - 1, pkg.max_words(w, b - a) + 2)
            out[fw:fw + len(words)] |= words
        return out, res, int(offs[-1]), int(ffo[-1])
    finally:
        for e in bands:
            e.close()


def test_bands_give_the_whole_images_stream(enc):
    """One stream from consecutive bands (what several GPUs do for one huge image): DC predictors from the
    block row above, global bit phase and 0xFF count handed from band to band."""
    pkg = load_pkg()
    rng = np.random.default_rng(44)
    noise = rng.integers(0, 256, (136, 200, 3), dtype=np.uint8)
    ones = np.full((48, 64, 3), 255, np.uint8)          # all-ones bytes straddle every cut
    ones[16:24, 8:40] = rng.integers(0, 256, (8, 32, 3), dtype=np.uint8)
    cases = [(natural_like(200, 328, 61), [0, 64, 72, 200]), (natural_like(200, 328, 62), [0, 200]),
             (noise, [0, 8, 16, 24, 136]), (noise, [0, 72, 136]), (ones, [0, 8, 16, 24, 32, 40, 48]),
             (natural_like(1080, 1920, 63), [0, 136, 544, 1080])]
    for k, (img, cuts) in enumerate(cases):
        want, r = O.encode(img)
        got, res, bits, ff = merged_bands(pkg, img, cuts, device_ptrs=(k % 2 == 1))
        assert bits == r.total_bits_unstuffed
        assert int(res["n_full_words"]) == r.n_full_words and int(res["tail_bits"]) == r.tail_bits
        assert int(res["total_bytes_stuffed"]) == r.total_bytes_stuffed and int(res["status"]) == 0
        assert np.array_equal(got[:len(want)], want), "case %d" % k
        assert not got[len(want):].any()
        # and the single-context path gives the same words
        words, _ = enc.encode(img)
        assert np.array_equal(words, want)


def test_edge_replication_extension(enc):
    """Sizes that are not multiples of 8 are rejected as the RTL would have it, unless the caller asks for
    the extension: then the stream is that of the image with its last row and column repeated."""
    pkg = load_pkg()
    rng = np.random.default_rng(91)
    for h, w in ((13, 21), (8, 9), (65, 64), (100, 203), (1, 1)):
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        with pytest.raises(pkg.JpegEncError) as err:
            enc.encode(rgb)
        assert err.value.code == pkg.E_BAD_ARG
        padded = np.pad(rgb, ((0, -h % 8), (0, -w % 8), (0, 0)), mode="edge")
        words, res = enc.encode(rgb, pad_edges=True)
        assert_same(words, res, padded)
    batch = rng.integers(0, 256, (3, 30, 50, 3), dtype=np.uint8)
    words, res = enc.encode_batch(batch, pad_edges=True)
    for i in range(3):
        assert_same(words[i], res[i], np.pad(batch[i], ((0, 2), (0, 6), (0, 0)), mode="edge"))
    # device pointers, strided rows
    torch = pytest.importorskip("torch")
    h, w, stride = 19, 45, 45 * 3 + 5
    buf = rng.integers(0, 256, (h, stride), dtype=np.uint8)
    rgb = np.ascontiguousarray(buf[:, :w * 3].reshape(h, w, 3))
    d_in = torch.from_numpy(buf).to("cuda:0")
    cap = pkg.max_words(w, h)
    d_words = torch.zeros(cap, dtype=torch.int32, device="cuda:0")
    d_res = torch.zeros(32, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    rc = enc.lib.jpegenc_encode_rgb8(enc._ctx, d_in.data_ptr(), w, h, stride, pkg.DEVICE_PTRS | pkg.PAD_EDGES,
                                     d_words.data_ptr(), cap, d_res.data_ptr())
    assert rc == 0
    enc.synchronize()
    r = d_res.cpu().numpy().view(pkg.RESULT_DTYPE)[0]
    k = int(r["n_full_words"]) + (1 if r["tail_bits"] else 0)
    assert_same(d_words.cpu().numpy().view(np.uint32)[:k], r, np.pad(rgb, ((0, 5), (0, 3), (0, 0)), mode="edge"))


def test_segment_calls_out_of_order_and_too_small():
    pkg = load_pkg()
    img = np.random.default_rng(71).integers(0, 256, (32, 64, 3), dtype=np.uint8)
    with pkg.JpegTop() as e:
        with pytest.raises(pkg.JpegEncError) as err:
            e.segment_count(0, 0)
        assert err.value.code == pkg.E_STATE
        bits, tail = e.segment_encode(img, 0, 32)
        with pytest.raises(pkg.JpegEncError) as err:
            e.segment_write(0, True, 64)  # the count step was left out
        assert err.value.code == pkg.E_STATE
        e.segment_count(0, 0)
        with pytest.raises(pkg.JpegEncError) as err:
            e.segment_write(0, True, 8)   # far too small: reported, nothing beyond 8 words touched
        assert err.value.code == pkg.E_TOO_SMALL
        # the context is usable afterwards
        words, res = e.encode(img)
        assert_same(words, res, img)
        assert int(res["total_bits_unstuffed"]) == bits


def test_output_too_small_is_reported_not_overrun(enc):
    pkg = load_pkg()
    rgb = np.random.default_rng(5).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    want, r = O.encode(rgb)
    cap = len(want) // 2
    lib = enc.lib
    words = np.full(cap + 16, 0xDEADBEEF, dtype=np.uint32)
    res = np.zeros(1, dtype=pkg.RESULT_DTYPE)
    rc = lib.jpegenc_encode_rgb8(enc._ctx, rgb.ctypes.data, 64, 64, 64 * 3, pkg.HOST_PTRS, words.ctypes.data, cap,
                                 res.ctypes.data)
    assert rc == pkg.E_TOO_SMALL
    assert int(res[0]["status"]) == pkg.E_TOO_SMALL
    assert int(res[0]["n_full_words"]) == r.n_full_words  # the size needed is still reported
    assert np.array_equal(words[:cap], want[:cap])
    assert (words[cap:] == 0xDEADBEEF).all()
    # bad dimensions
    assert lib.jpegenc_encode_rgb8(enc._ctx, rgb.ctypes.data, 60, 64, 180, 0, words.ctypes.data, cap,
                                   res.ctypes.data) == pkg.E_BAD_ARG
    assert lib.jpegenc_encode_rgb8(enc._ctx, rgb.ctypes.data, 64, 64, 100, 0, words.ctypes.data, cap,
                                   res.ctypes.data) == pkg.E_BAD_ARG


def test_device_pointers_strides_and_alignment():
    torch = pytest.importorskip("torch")
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(13)
    h, w, n = 40, 104, 3
    imgs = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    cap = pkg.max_words(w, h)
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        for pad, shift in ((0, 0), (8, 0), (4, 4), (5, 1), (3, 3)):
            stride = w * 3 + pad
            istride = stride * h + (16 if shift == 0 else 7)
            buf = np.zeros(shift + istride * n + 64, dtype=np.uint8)
            for i in range(n):
                for r in range(h):
                    o = shift + i * istride + r * stride
                    buf[o:o + w * 3] = imgs[i, r].reshape(-1)
            d_in = torch.from_numpy(buf).to(dev)
            d_words = torch.zeros((n, cap), dtype=torch.int32, device=dev)
            d_res = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
            e.encode_batch_device(d_in.data_ptr() + shift, n, w, h, stride, istride, d_words.data_ptr(), cap,
                                  d_res.data_ptr())
            torch.cuda.synchronize()
            res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(n)
            words = d_words.cpu().numpy().view(np.uint32)
            for i in range(n):
                k = int(res[i]["n_full_words"]) + (1 if res[i]["tail_bits"] else 0)
                assert_same(words[i, :k], res[i], imgs[i])


def test_row_strides_around_the_32_bit_offset_limit():
    """The 8-byte-aligned kernel variant addresses an image with 32-bit offsets from its first byte and is
    used while rows * stride < 2^31: a 2 GiB device buffer holds a 16-row image at a stride just below that
    bound (largest offsets the variant sees) and at the bound (the 64-bit-address variant takes over)."""
    torch = pytest.importorskip("torch")
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    h, w = 16, 192
    img = np.random.default_rng(77).integers(0, 256, (h, w, 3), dtype=np.uint8)
    d_img = torch.from_numpy(img.reshape(h, w * 3)).to(dev)
    cap = pkg.max_words(w, h)
    d_buf = torch.zeros((1 << 31) + 4096, dtype=torch.uint8, device=dev)
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        for stride in ((1 << 27) - 8, 1 << 27):
            d_buf.zero_()
            for r in range(h):
                d_buf[r * stride:r * stride + w * 3] = d_img[r]
            d_words = torch.zeros((1, cap), dtype=torch.int32, device=dev)
            d_res = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
            e.encode_batch_device(d_buf.data_ptr(), 1, w, h, stride, stride * h, d_words.data_ptr(), cap,
                                  d_res.data_ptr())
            torch.cuda.synchronize()
            res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(1)
            words = d_words.cpu().numpy().view(np.uint32)
            k = int(res[0]["n_full_words"]) + (1 if res[0]["tail_bits"] else 0)
            assert_same(words[0, :k], res[0], img)
    del d_buf


def test_4k_batch_properties():
    """Config-2 shape (3840x2160 frames, device resident) at a bounded count: two frames are checked
    word for word against the oracle, the rest through size-independent properties -- equal images give
    equal streams wherever they sit in the batch, no 0xFF byte is left unstuffed, and the bit count
    equals the oracle's coefficient-domain count."""
    torch = pytest.importorskip("torch")
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    h, w, n = 2160, 3840, 6
    base = [natural_like(h, w, 100), np.random.default_rng(101).integers(0, 256, (h, w, 3), dtype=np.uint8)]
    order = [0, 1, 0, 0, 1, 0]
    d_in = torch.stack([torch.from_numpy(base[k]) for k in order]).to(dev)
    cap = pkg.max_words(w, h) // 3
    d_words = torch.zeros((n, cap), dtype=torch.int32, device=dev)
    d_res = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        e.encode_batch_device(d_in.data_ptr(), n, w, h, w * 3, h * w * 3, d_words.data_ptr(), cap, d_res.data_ptr())
        torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(n)
    words = d_words.cpu().numpy().view(np.uint32)
    for k in (0, 1):
        i = order.index(k)
        cnt = int(res[i]["n_full_words"]) + (1 if res[i]["tail_bits"] else 0)
        assert_same(words[i, :cnt], res[i], base[k])
        by = words[i, :cnt].astype(">u4").tobytes()[:int(res[i]["total_bytes_stuffed"])]
        a = np.frombuffer(by, dtype=np.uint8)
        ff = np.nonzero(a[:-1] == 0xFF)[0]
        assert (a[ff + 1] == 0).all()
    for i, k in enumerate(order):
        j = order.index(k)
        assert res[i].tobytes() == res[j].tobytes()
        cnt = int(res[i]["n_full_words"]) + 1
        assert np.array_equal(words[i, :cnt], words[j, :cnt])
