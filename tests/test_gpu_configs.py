"""GPU tests at the sizes of the named configurations: a device-resident batch of 1080p frames (the
per-GPU share of the sharded 8192-frame configuration), a 16384 x 16384 image as eight restart-interval
stripes, the multi-GPU glue driven by the HIP path (one rank: the exchange degenerates, the code path is
the one every rank runs), two streams sharing one context, and the HIP stream inside a JFIF container.
Every comparison is word for word against the CPU oracle unless it says otherwise."""
import io
import socket
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle_lib as O
from conftest import load_pkg, natural_like

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def same_as_oracle(words, res, rgb, cfg=None):
    want, r = O.encode(rgb, cfg)
    got = (int(res["n_full_words"]), int(res["tail_bits"]), int(res["tail_word"]), int(res["total_bits_unstuffed"]),
           int(res["total_bytes_stuffed"]), int(res["status"]))
    assert got == (r.n_full_words, r.tail_bits, r.tail_word, r.total_bits_unstuffed, r.total_bytes_stuffed, 0)
    assert len(words) == len(want) and np.array_equal(words, want)


def no_unstuffed_ff(words, res):
    a = np.frombuffer(np.asarray(words, dtype=np.uint32).astype(">u4").tobytes()[:int(res["total_bytes_stuffed"])],
                      dtype=np.uint8)
    ff = np.nonzero(a[:-1] == 0xFF)[0]
    return bool((a[ff + 1] == 0).all())


def test_1080p_batch_of_64_distinct_frames():
    """64 device-resident 1080p frames in one jpegenc_encode_batch call: 60 distinct natural-like frames and
    4 repeats.  Ten frames word for word against the oracle; all of them by the properties that do not need
    it (status, stuffing, repeats give identical streams wherever they sit in the batch)."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    h, w, n = 1080, 1920, 64
    frames = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
    for s in range(0, 60, 12):
        frames[s:s + 12] = pkg.synth.natural_like_torch(12, h, w, 4000 + s, dev)
    frames[60:64] = frames[torch.tensor([0, 17, 33, 59], device=dev)]
    cap = w * h * 3 // 4
    d_words = torch.zeros((n, cap), dtype=torch.int32, device=dev)
    d_res = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        e.reserve(n, w, h)
        e.encode_batch_device(frames.data_ptr(), n, w, h, w * 3, h * w * 3, d_words.data_ptr(), cap, d_res.data_ptr())
        torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(n)
    assert (res["status"] == 0).all()
    counts = res["n_full_words"].astype(np.int64) + (res["tail_bits"] != 0)
    checked = [0, 1, 11, 12, 31, 32, 47, 59, 62, 63]
    host = {i: frames[i].cpu().numpy() for i in checked}
    got = {i: d_words[i, :int(counts[i])].cpu().numpy().view(np.uint32) for i in range(n)}
    with ThreadPoolExecutor(8) as pool:  # the oracle releases the GIL inside its C call
        list(pool.map(lambda i: same_as_oracle(got[i], res[i], host[i]), checked))
    for i in range(n):
        assert no_unstuffed_ff(got[i], res[i])
    for i, j in zip((60, 61, 62, 63), (0, 17, 33, 59)):
        assert res[i].tobytes() == res[j].tobytes() and np.array_equal(got[i], got[j])
    sizes = {int(c) for c in counts[:60]}
    assert len(sizes) > 50  # distinct frames really were distinct


def test_16384_square_image_as_eight_stripes():
    """A 16384 x 16384 image through jpegenc_encode_stripes: every one of the eight 16384 x 2048 stripes must be
    the stream jpeg_top gives for that sub-image after rst."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    n, stripes = 16384, 8
    rows = n // stripes
    img = torch.cat([pkg.synth.natural_like_torch(1, rows, n, 7 + k, dev)[0] for k in range(stripes)])
    # a block of pure noise across the boundary of stripes 2 and 3: long strings and 0xFF bytes at a restart
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    img[3 * rows - 64:3 * rows + 64, 1000:3000] = torch.randint(0, 256, (128, 2000, 3), generator=g, device=dev,
                                                                 dtype=torch.uint8)
    cap = n * rows * 3 // 4
    d_words = torch.zeros((stripes, cap), dtype=torch.int32, device=dev)
    d_res = torch.zeros((stripes, 32), dtype=torch.uint8, device=dev)
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        e.encode_stripes_device(img.data_ptr(), n, n, n * 3, stripes, d_words.data_ptr(), cap, d_res.data_ptr())
        torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(stripes)
    assert (res["status"] == 0).all()
    host = img.cpu().numpy()
    del img

    def check(s):
        k = int(res[s]["n_full_words"]) + (1 if res[s]["tail_bits"] else 0)
        same_as_oracle(d_words[s, :k].cpu().numpy().view(np.uint32), res[s], host[s * rows:(s + 1) * rows])

    with ThreadPoolExecutor(8) as pool:
        list(pool.map(check, range(stripes)))


@pytest.fixture
def one_rank_group():
    """A process group of one rank on the GPU backend (RCCL); the functions under test run the code every
    rank of a multi-GPU job runs, with the exchange reduced to rank 0 talking to itself."""
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_stripe_gather_and_band_merge_fed_by_the_hip_path(one_rank_group):
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(12)
    img = natural_like(512, 1024, 21)
    img[200:280, 300:700] = rng.integers(0, 256, (80, 400, 3), dtype=np.uint8)
    d_img = torch.from_numpy(img).to(dev)
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        # restart-interval stripes: table + payload on the root
        n_stripes = 8
        table, payload = pkg.distributed.encode_striped_image(pkg.distributed.hip_stripe_coder(e, d_img, n_stripes),
                                                              n_stripes, device=dev)
        payload = payload.cpu().numpy().view(np.uint32)
        off = 0
        for s in range(n_stripes):
            want, r = O.encode(img[64 * s:64 * (s + 1)])
            assert table[s, 0] == off
            assert tuple(table[s, 1:4]) == (r.n_full_words, r.tail_bits, r.tail_word)
            assert table[s, 6] == len(want) and np.array_equal(payload[off:off + len(want)], want)
            off += len(want)
        assert off == len(payload)
        # the same with all stripes of the rank coded by one call and their records gathered unread
        table2, payload2 = pkg.distributed.encode_striped_image(pkg.distributed.HipStripes(e, d_img, n_stripes), n_stripes,
                                                                device=dev)
        assert np.array_equal(table2, table) and np.array_equal(payload2.cpu().numpy().view(np.uint32), payload)
        # one stream: the band coder is the HIP segment path, the words never leave the GPU before the merge
        lo, hi = pkg.distributed.band_rows(img.shape[0], 0, 1)
        want, r = O.encode(img)
        for coder in (pkg.distributed.HipBand, pkg.distributed.HipBandResident):
            words, res = pkg.distributed.encode_monolithic(coder(e, d_img, lo, hi), device=dev)
            words = words.cpu().numpy().view(np.uint32)
            assert (res["n_full_words"], res["tail_bits"], res["tail_word"]) == (r.n_full_words, r.tail_bits, r.tail_word)
            assert np.array_equal(words[:len(want)], want) and not words[len(want):].any()


def test_bands_with_their_figures_left_on_the_gpu():
    """The device-resident segment calls: several bands of one image, each on a context of its own, offsets summed
    by torch on the GPU from the figures the calls leave there -- no copy to the host and no wait until the spans
    are read at the end; the ORed words are the whole image's stream."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    for shape, cuts in (((96, 200, 3), [0, 8, 40, 96]), ((512, 1024, 3), [0, 64, 256, 264, 512]), ((24, 8, 3), [0, 8, 16, 24])):
        img = natural_like(shape[0], shape[1], 40 + shape[0])
        img[shape[0] // 3:shape[0] // 2] = rng.integers(0, 256, (shape[0] // 2 - shape[0] // 3, shape[1], 3), dtype=np.uint8)
        if shape[1] == 8:
            img[8:] = img[8, 0]  # a last band of 18 bits inside the word the band before it ends in
        d_img = torch.from_numpy(img).to(dev)
        want, r = O.encode(img)
        encs = [pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) for _ in cuts[1:]]
        try:
            bands = [pkg.distributed.HipBandResident(e, d_img, a, b) for e, a, b in zip(encs, cuts[:-1], cuts[1:])]
            rows = torch.stack([b.encode() for b in bands])
            zero = torch.zeros(1, dtype=torch.int64, device=dev)
            ffs = torch.cat([b.count(rows[:i, 0].sum().reshape(1), rows[i - 1, 1].reshape(1) if i else zero)
                             for i, b in enumerate(bands)])
            outs = [b.write(ffs[:i].sum().reshape(1), i == len(bands) - 1) for i, b in enumerate(bands)]
            merged = torch.zeros(len(want) + 8, dtype=torch.int32, device=dev)
            spans = torch.stack([o[0] for o in outs]).cpu().numpy()  # the first wait
            for (span, rec, words), (fw, n) in zip(outs, spans):
                assert n <= words.numel()
                merged[int(fw):int(fw) + int(n)] |= words[:int(n)]
            got = merged.cpu().numpy().view(np.uint32)
            rec = outs[-1][1].cpu().numpy().view(pkg.RESULT_DTYPE)[0]
            assert np.array_equal(got[:len(want)], want) and not got[len(want):].any()
            assert (int(rec["n_full_words"]), int(rec["tail_bits"]), int(rec["total_bytes_stuffed"]), int(rec["status"])) == \
                   (r.n_full_words, r.tail_bits, r.total_bytes_stuffed, 0)
        finally:
            for e in encs:
                e.close()


def test_two_streams_share_one_context():
    """Device-pointer calls only enqueue work and all calls of a context share its workspace: when the caller
    moves the context to another stream, the batch still running on the first stream must finish with the
    workspace before the second batch overwrites it."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    h, w, n = 1080, 1920, 24
    a = pkg.synth.natural_like_torch(n, h, w, 900, dev)
    b = pkg.synth.natural_like_torch(n, h, w, 901, dev)
    cap = w * h * 3 // 4
    out = [torch.zeros((n, cap), dtype=torch.int32, device=dev) for _ in range(2)]
    res = [torch.zeros((n, 32), dtype=torch.uint8, device=dev) for _ in range(2)]
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    with pkg.JpegTop() as e:
        e.reserve(n, w, h)
        for rep in range(3):  # back to back, never a host synchronisation in between
            e.set_stream(s1.cuda_stream)
            e.encode_batch_device(a.data_ptr(), n, w, h, w * 3, h * w * 3, out[0].data_ptr(), cap, res[0].data_ptr())
            e.set_stream(s2.cuda_stream)
            e.encode_batch_device(b.data_ptr(), n, w, h, w * 3, h * w * 3, out[1].data_ptr(), cap, res[1].data_ptr())
        torch.cuda.synchronize()
    for frames, o, r in ((a, out[0], res[0]), (b, out[1], res[1])):
        rr = r.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(n)
        assert (rr["status"] == 0).all()
        for i in (0, n // 2, n - 1):
            k = int(rr[i]["n_full_words"]) + (1 if rr[i]["tail_bits"] else 0)
            same_as_oracle(o[i, :k].cpu().numpy().view(np.uint32), rr[i], frames[i].cpu().numpy())
        # the frames between them: equal to a second encode done alone
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        for frames, o in ((a, out[0]), (b, out[1])):
            again = torch.zeros_like(o)
            r2 = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
            e.encode_batch_device(frames.data_ptr(), n, w, h, w * 3, h * w * 3, again.data_ptr(), cap, r2.data_ptr())
            torch.cuda.synchronize()
            rr = r2.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(n)
            for i in range(n):
                k = int(rr[i]["n_full_words"]) + (1 if rr[i]["tail_bits"] else 0)
                assert torch.equal(again[i, :k], o[i, :k])


def test_hip_stream_in_a_jfif_container_decodes():
    """The GPU's words wrapped in JFIF headers decode with libjpeg (Pillow) to the input: default tables
    nearly lossless, a quantised table at its expected quality, an odd size through the padding extension."""
    Image = pytest.importorskip("PIL.Image")
    pkg = load_pkg()

    def psnr(x, y):
        mse = np.mean((x.astype(np.float64) - y.astype(np.float64)) ** 2)
        return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)

    def decode(jpg):
        return np.asarray(Image.open(io.BytesIO(jpg)).convert("RGB"))

    rgb = natural_like(136, 200, 4)
    with pkg.JpegTop() as e:
        words, res = e.encode(rgb)
        out = decode(pkg.jfif.wrap_jfif(words, int(res["n_full_words"]), int(res["tail_bits"]), 200, 136))
        assert out.shape == rgb.shape and psnr(out, rgb) > 40.0
        odd = rgb[:101, :77]
        words, res = e.encode(odd, pad_edges=True)
        out = decode(pkg.jfif.wrap_jfif(words, int(res["n_full_words"]), int(res["tail_bits"]), 77, 101))
        assert out.shape == odd.shape and psnr(out, odd) > 40.0
    qt = [np.full(64, 8), np.full(64, 16), np.full(64, 16)]
    with pkg.JpegTop(qtabs=qt) as e:
        words, res = e.encode(rgb)
        out = decode(pkg.jfif.wrap_jfif(words, int(res["n_full_words"]), int(res["tail_bits"]), 200, 136, qtabs=qt))
        assert psnr(out, rgb) > 28.0
    conv = np.array([[32 if b >= 3 else 2 for b in range(8)] for a in range(8)]).reshape(64)
    with pkg.JpegTop(qtabs=[conv] * 3, quant_index_order=1) as e:
        words, res = e.encode(rgb)
        good = decode(pkg.jfif.wrap_jfif(words, int(res["n_full_words"]), int(res["tail_bits"]), 200, 136,
                                         qtabs=[conv] * 3, quant_index_order=1))
        assert psnr(good, rgb) > 28.0


def test_segment_state_is_dropped_by_other_calls_and_one_chunk_last_band():
    """An encode call between the segment steps discards the band's chunk streams: the later steps must say
    so.  And a last band of a single chunk whose first word begins in the band before it reports as
    tail_word its own bytes only (the neighbour's read as zero)."""
    pkg = load_pkg()
    img = np.random.default_rng(72).integers(0, 256, (32, 64, 3), dtype=np.uint8)
    with pkg.JpegTop() as e:
        e.segment_encode(img, 0, 32)
        e.encode(img)
        with pytest.raises(pkg.JpegEncError) as err:
            e.segment_count(0, 0)
        assert err.value.code == pkg.E_STATE
        e.segment_encode(img, 0, 32)
        e.segment_count(0, 0)
        e.encode(img)
        with pytest.raises(pkg.JpegEncError) as err:
            e.segment_write(0, True, 4096)
        assert err.value.code == pkg.E_STATE
    # an image one MCU wide, bands of one MCU: every band is a single chunk, and the flat last two MCUs make the
    # last band 18 bits that mostly sit inside the word the band before it ends in
    for seed in range(8):
        img = natural_like(24, 8, 300 + seed)
        img[8:] = img[8, 0]
        want, r = O.encode(img)
        bands = [pkg.JpegTop() for _ in range(3)]
        try:
            info = [b.segment_encode(img, 8 * i, 8) for i, b in enumerate(bands)]
            offs = np.concatenate([[0], np.cumsum([i[0] for i in info])])
            ffs = [b.segment_count(int(offs[i]), info[i - 1][1] if i else 0) for i, b in enumerate(bands)]
            ffo = np.concatenate([[0], np.cumsum(ffs)])
            merged = np.zeros(len(want) + 4, dtype=np.uint32)
            parts = []
            for i, b in enumerate(bands):
                fw, words, res = b.segment_write(int(ffo[i]), i == 2, 4096)
                merged[fw:fw + len(words)] |= words
                parts.append((fw, words))
            assert np.array_equal(merged[:len(want)], want)
            if int(res["tail_bits"]):
                fw, words = parts[-1]
                assert int(res["tail_word"]) == int(words[int(res["n_full_words"]) - fw])  # its own bytes only
        finally:
            for b in bands:
                b.close()


def test_region_modes_give_the_same_stream():
    """The encode kernel exists with short, small and large LDS staging regions (and a fourth size in the
    single-launch form); the context picks one per launch from the overflow counts of the launches before.
    Whatever runs, the words are the oracle's: easy, mixed (every other MCU row noise) and noise content
    under the three forced sizes and in the single-launch form, and a run in automatic mode that starts on easy
    content (and moves to short regions if no block of it is too long for them), meets the others (leaves short
    regions at once, switches to large ones) and returns to easy content.  The overflow counts the choice rests on
    are checked too: none on easy content, every chunk on noise."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    h, w, n = 256, 1024, 36  # 36 x 64 chunks per call: beyond what the single-launch form takes
    g = torch.Generator(device=dev)
    g.manual_seed(31)
    easy = pkg.synth.natural_like_torch(n, h, w, 40, dev)
    noise = torch.randint(0, 256, (n, h, w, 3), generator=g, device=dev, dtype=torch.uint8)
    mixed = easy.clone()
    mixed[:, 8::16] = noise[:, 8::16]  # rows 8, 24, ...: one pixel row of noise in every other MCU row
    mixed[:, 12::16] = noise[:, 12::16]
    # pixels of 0 and 255 only: the longest codes pixels can give, four to five lane ranges per chunk with short regions
    binary = (torch.randint(0, 2, (n, h, w, 3), generator=g, device=dev, dtype=torch.uint8) * 255).contiguous()
    cap = pkg.max_words(w, h)
    want = {}

    def run(e, name, frames, count=None):
        k_img = frames.shape[0] if count is None else count
        d_words = torch.zeros((k_img, cap), dtype=torch.int32, device=dev)
        d_res = torch.zeros((k_img, 32), dtype=torch.uint8, device=dev)
        e.encode_batch_device(frames.data_ptr(), k_img, w, h, w * 3, h * w * 3, d_words.data_ptr(), cap, d_res.data_ptr())
        torch.cuda.synchronize()
        res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(k_img)
        assert (res["status"] == 0).all()
        if name not in want:  # the oracle's words, once per content
            host = frames.cpu().numpy()
            want[name] = [O.encode(host[i])[0] for i in range(n)]
        for i in range(k_img):
            k = int(res[i]["n_full_words"]) + (1 if res[i]["tail_bits"] else 0)
            assert np.array_equal(d_words[i, :k].cpu().numpy().view(np.uint32), want[name][i]), (name, i)

    contents = (("easy", easy), ("mixed", mixed), ("noise", noise))
    J = pkg.JpegTop
    easy_over_short = None
    for mode in (J.REGIONS_SMALL, J.REGIONS_LARGE, J.REGIONS_SHORT):
        with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
            e.set_region_mode(mode)
            for name, frames in contents:
                run(e, name, frames)
                over_small, over_large, total, over_short, stands_at = e.region_stats()
                assert total == n * 64 and over_large <= over_small <= over_short <= total
                assert stands_at == mode
                if name == "easy":
                    assert over_small == 0
                    assert easy_over_short in (None, over_short)  # the same count whichever variant counted
                    easy_over_short = over_short
                if name == "noise":
                    assert over_small == total
            run(e, "binary", binary)
            assert e.region_stats()[1] > 0  # (some blocks are too long even for a large region)
            e.set_pipeline(pkg.JpegTop.PIPELINE_SPLIT)  # a small call as separate kernels takes the forced size too
            for name, frames in contents:
                run(e, name, frames, 3)
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:  # the single-launch form
        for name, frames in contents:
            run(e, name, frames, 30)
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:  # automatic
        for bad in (-1, 4):
            with pytest.raises(pkg.JpegEncError):
                e.set_region_mode(bad)
        assert e.region_stats()[4] == J.REGIONS_SMALL
        for _ in range(6):
            run(e, "easy", easy)
        if easy_over_short * 1024 < n * 64:  # four launches' counts in favour: short regions from the fifth on
            assert e.region_stats()[4] == J.REGIONS_SHORT
        run(e, "mixed", mixed)  # (coded with whatever the easy launches left; its counts arrive behind it)
        run(e, "mixed", mixed)
        assert e.region_stats()[4] != J.REGIONS_SHORT
        for name, frames in (contents[1], contents[2], contents[2]):
            run(e, name, frames)
        assert e.region_stats()[4] == J.REGIONS_LARGE
        for name, frames in (contents[0], contents[0], contents[0]):
            run(e, name, frames)


def test_one_image_per_compute_unit_takes_the_sixteen_wave_stuffing_kernel():
    """256 to 511 images that fill the compute units in whole turns, of at least 128 chunks each, take the per-image
    stuffing kernel with 16 waves per workgroup (no barrier per round: every chunk's 0xFF count has a word in the
    LDS).  256 images of 128 chunks of mixed content: word for word against the same batch coded in two halves
    (per-chunk kernels), every eighth image against the oracle."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    h, w, n = 512, 1024, n_cu
    rng = np.random.default_rng(256)
    imgs = np.empty((n, h, w, 3), np.uint8)
    base = [natural_like(h, w, 3000 + k) for k in range(4)]
    for i in range(n):
        kind = i % 4
        imgs[i] = base[kind]
        if kind == 1:
            imgs[i, 100:164] = rng.integers(0, 256, (64, w, 3), dtype=np.uint8)  # a band of noise: long chunks
        elif kind == 2:
            imgs[i, :, 512:] = 255  # all-ones bytes
        elif kind == 3:
            imgs[i, 8 * (i % 60):8 * (i % 60) + 8] = rng.integers(0, 256, 3, dtype=np.uint8)
    d_in = torch.from_numpy(imgs).to(dev)
    cap = pkg.max_words(w, h) // 2
    outs = []
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        e.set_profiling(True)
        for a, b in ((0, n), (0, n // 2), (n // 2, n)):
            d_words = torch.zeros((b - a, cap), dtype=torch.int32, device=dev)
            d_res = torch.zeros((b - a, 32), dtype=torch.uint8, device=dev)
            e.encode_batch_device(d_in[a:b].data_ptr(), b - a, w, h, w * 3, h * w * 3, d_words.data_ptr(), cap,
                                  d_res.data_ptr())
            torch.cuda.synchronize()
            times = e.kernel_times(reset=True)
            assert (times["k_stuff_image"][0] == 1) == (b - a == n) and (times["k_stuff_write"][0] == 1) == (b - a < n)
            outs.append((d_words.cpu().numpy().view(np.uint32), d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(b - a)))
    words, res = outs[0]
    assert all(int(r["status"]) == 0 for r in res)
    assert res.tobytes() == np.concatenate([outs[1][1], outs[2][1]]).tobytes()
    assert np.array_equal(words, np.concatenate([outs[1][0], outs[2][0]]))
    for i in range(0, n, 8):
        want, r = O.encode(imgs[i])
        assert int(res[i]["n_full_words"]) == r.n_full_words and int(res[i]["tail_bits"]) == r.tail_bits
        assert int(res[i]["tail_word"]) == r.tail_word and np.array_equal(words[i, :len(want)], want)


def test_many_images_take_the_per_image_stuffing_kernel():
    """From 512 images of at least 32 chunks on, one launch with one workgroup per image does both stuffing steps
    (k_stuff_image).  520 images of 34 chunks (the last one partial) of every kind of content -- natural, noise
    (chunk streams of more than five 1 KB steps), flat (the shortest possible chunks), all-ones bytes around
    noise -- every one word for word against the oracle, and against the same batch coded in two halves (which
    takes the per-chunk kernels)."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    h, w, n = 264, 520, 520
    rng = np.random.default_rng(130)
    imgs = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        kind = i % 5
        if kind == 0:
            imgs[i] = natural_like(h, w, 2000 + i % 40)
        elif kind == 1:
            imgs[i] = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        elif kind == 2:
            imgs[i] = rng.integers(0, 256, 3, dtype=np.uint8)  # flat
        elif kind == 3:
            imgs[i] = 255
            imgs[i, 40:200, 100:400] = rng.integers(0, 256, (160, 300, 3), dtype=np.uint8)
        else:
            imgs[i] = imgs[i - 4]
            imgs[i, :, 256:] = rng.choice(np.array([0, 255], dtype=np.uint8), size=(h, w - 256, 3))
    d_in = torch.from_numpy(imgs).to(dev)
    cap = pkg.max_words(w, h)
    outs = []
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        e.set_profiling(True)
        for parts in ((0, n), (0, 260), (260, n)):
            a, b = parts
            d_words = torch.zeros((b - a, cap), dtype=torch.int32, device=dev)
            d_res = torch.zeros((b - a, 32), dtype=torch.uint8, device=dev)
            e.encode_batch_device(d_in[a:b].data_ptr(), b - a, w, h, w * 3, h * w * 3, d_words.data_ptr(), cap,
                                  d_res.data_ptr())
            torch.cuda.synchronize()
            times = e.kernel_times(reset=True)
            assert (times["k_stuff_image"][0] == 1) == (b - a >= 512) and (times["k_stuff_write"][0] == 1) == (b - a < 512)
            outs.append((d_words.cpu().numpy().view(np.uint32), d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(b - a)))
    words, res = outs[0]
    halves_w = np.concatenate([outs[1][0], outs[2][0]])
    halves_r = np.concatenate([outs[1][1], outs[2][1]])
    assert res.tobytes() == halves_r.tobytes() and np.array_equal(words, halves_w)

    def check(i):
        k = int(res[i]["n_full_words"]) + (1 if res[i]["tail_bits"] else 0)
        same_as_oracle(words[i, :k], res[i], imgs[i])

    with ThreadPoolExecutor(8) as pool:
        list(pool.map(check, range(n)))


def test_single_launch_calls_of_two_contexts_share_the_gpu():
    """Two contexts on two streams, each queueing calls of four 1080p frames (2028 chunks: the single-launch
    form at its limit).  Both launches cannot be resident together, so chunks of one wait for CUs while chunks of
    the other wait for their predecessors' hand-off words: everything must still come out right, every time."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    h, w, n, reps = 1080, 1920, 4, 40
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    a = pkg.synth.natural_like_torch(n, h, w, 910, dev)
    b = torch.randint(0, 256, (n, h, w, 3), generator=g, device=dev, dtype=torch.uint8)
    b[:, :, :960] = pkg.synth.natural_like_torch(n, h, 960, 911, dev)  # half noise: chunks of very different duration
    cap = w * h * 6 // 4
    want = {}
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        e.set_pipeline(e.PIPELINE_SPLIT)
        for name, frames in (("a", a), ("b", b)):
            d_words = torch.zeros((n, cap), dtype=torch.int32, device=dev)
            d_res = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
            e.encode_batch_device(frames.data_ptr(), n, w, h, w * 3, h * w * 3, d_words.data_ptr(), cap, d_res.data_ptr())
            torch.cuda.synchronize()
            want[name] = (d_words, d_res)
    res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(n)
    k = int(res[0]["n_full_words"]) + (1 if res[0]["tail_bits"] else 0)
    same_as_oracle(want["b"][0][0, :k].cpu().numpy().view(np.uint32), res[0], b[0].cpu().numpy())
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    outs = {"a": [], "b": []}
    torch.cuda.synchronize()
    with pkg.JpegTop(stream=s1.cuda_stream) as ea, pkg.JpegTop(stream=s2.cuda_stream) as eb:
        for rep in range(reps):
            for name, e, frames in (("a", ea, a), ("b", eb, b)):
                d_words = torch.zeros((n, cap), dtype=torch.int32, device=dev)
                d_res = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
                outs[name].append((d_words, d_res))
            torch.cuda.synchronize()  # the buffers are zeroed before the two streams start on them
            for name, e, frames in (("a", ea, a), ("b", eb, b)):
                d_words, d_res = outs[name][-1]
                e.encode_batch_device(frames.data_ptr(), n, w, h, w * 3, h * w * 3, d_words.data_ptr(), cap, d_res.data_ptr())
            if rep % 8 == 7:
                torch.cuda.synchronize()
                for name in ("a", "b"):
                    for d_words, d_res in outs[name]:
                        assert torch.equal(d_res, want[name][1]) and torch.equal(d_words, want[name][0])
                    outs[name].clear()
        torch.cuda.synchronize()


def test_slots_sized_from_experience():
    """JPEGENC_SLOTS_ADAPTIVE: same words as with worst-case slots from a fraction of the workspace; content whose
    chunks outgrow the slots comes back flagged per image (device pointers) and fits when the call is repeated, and
    is repeated by the library itself for host pointers."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    h, w, n = 1080, 1920, 24  # 24 x 507 chunks: a large call
    natural = pkg.synth.natural_like_torch(n, h, w, 4100, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    noise = torch.randint(0, 256, (n, h, w, 3), generator=g, device=dev, dtype=torch.uint8)
    mixed = natural.clone()
    mixed[5] = noise[5]
    cap = w * h * 6 // 4

    def run(e, frames):
        d_words = torch.zeros((n, cap), dtype=torch.int32, device=dev)
        d_res = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
        e.encode_batch_device(frames.data_ptr(), n, w, h, w * 3, h * w * 3, d_words.data_ptr(), cap, d_res.data_ptr())
        torch.cuda.synchronize()
        return d_words, d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(n)

    def same(a, ra, b, rb, which):
        for i in which:
            assert ra[i].tobytes() == rb[i].tobytes(), i
            k = int(ra[i]["n_full_words"]) + 1
            assert torch.equal(a[i, :k], b[i, :k]), i

    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as ref, \
            pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        e.set_slot_mode(e.SLOTS_ADAPTIVE)
        want_nat, rn = run(ref, natural)
        want_noise, rz = run(ref, noise)
        want_mixed, rm = run(ref, mixed)
        assert (rn["status"] == 0).all() and (rz["status"] == 0).all() and (rm["status"] == 0).all()
        worst_bytes = ref.workspace_bytes()
        for _ in range(2):
            got, r = run(e, natural)
            same(got, r, want_nat, rn, range(n))
        assert e.workspace_bytes() * 2 < worst_bytes  # (the first nine frames went through with worst-case slots)
        # one frame of noise among the photographs: that image is flagged, the others are right
        got, r = run(e, mixed)
        assert int(r[5]["status"]) == pkg.E_AGAIN and (np.delete(r["status"], 5) == 0).all()
        same(got, r, want_mixed, rm, [i for i in range(n) if i != 5])
        got, r = run(e, mixed)  # repeated: the context has seen the long chunks
        same(got, r, want_mixed, rm, range(n))
        got, r = run(e, noise)
        same(got, r, want_noise, rz, range(n))
        # and back: the slots stay large for a while, the output is the same
        got, r = run(e, natural)
        same(got, r, want_nat, rn, range(n))
    # host pointers: the library repeats the call itself
    with pkg.JpegTop() as e:
        e.set_slot_mode(e.SLOTS_ADAPTIVE)
        host_nat, host_noise = natural[:6].cpu().numpy(), noise[:6].cpu().numpy()
        words, res = e.encode_batch(host_nat)
        assert (res["status"] == 0).all()
        words, res = e.encode_batch(host_noise)
        assert (res["status"] == 0).all()
        for i in (0, 5):
            k = int(rz[i]["n_full_words"]) + (1 if rz[i]["tail_bits"] else 0)
            assert np.array_equal(words[i], want_noise[i, :k].cpu().numpy().view(np.uint32))
        one, r = O.encode(host_noise[3])
        assert np.array_equal(words[3], one)


def test_256_frames_of_3840x2160_in_one_call():
    """The 256 x 3840x2160 configuration at its full count, device resident, one call (one 16-wave stuffing workgroup
    per image): four frames word for word against the oracle, every frame through what holds at any size -- status,
    no 0xFF byte left unstuffed, sizes consistent with each other, equal images giving equal streams wherever they
    sit in the batch."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    h, w, n = 2160, 3840, 256
    frames = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
    for s in range(0, n, 8):
        frames[s:s + 8] = pkg.synth.natural_like_torch(8, h, w, 100 + s, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    frames[77, 500:900, 1000:2600] = torch.randint(0, 256, (400, 1600, 3), generator=g, device=dev, dtype=torch.uint8)
    twins = {130: 5, 255: 77, 201: 5}  # copies of earlier frames, the noisy one among them
    for dst, src in twins.items():
        frames[dst] = frames[src]
    cap = w * h * 3 // 4
    d_words = torch.zeros((n, cap), dtype=torch.int32, device=dev)
    d_res = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        e.encode_batch_device(frames.data_ptr(), n, w, h, w * 3, h * w * 3, d_words.data_ptr(), cap, d_res.data_ptr())
        torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(n)
    assert (res["status"] == 0).all()
    # sizes: stuffed bytes = whole unstuffed bytes + stuffed zeros >= the former; words and tail from the stuffed bits
    sbits = res["total_bytes_stuffed"].astype(np.int64) * 8 + (res["total_bits_unstuffed"].astype(np.int64) & 7)
    assert (res["total_bytes_stuffed"] >= res["total_bits_unstuffed"] // 8).all()
    assert (res["n_full_words"] == sbits >> 5).all() and (res["tail_bits"] == (sbits & 31)).all()
    assert (res["eof_partial_ready"] == (res["tail_bits"] != 0)).all()
    for i in range(n):  # stuffing, on the GPU: every 0xFF byte of the stream is followed by 0x00
        k = int(res[i]["total_bytes_stuffed"])
        wv = d_words[i, :(k + 3) // 4]
        by = torch.stack([(wv >> 24) & 0xFF, (wv >> 16) & 0xFF, (wv >> 8) & 0xFF, wv & 0xFF], dim=1).reshape(-1)[:k]
        ff = by[:-1] == 0xFF
        assert bool((by[1:][ff] == 0).all()) and int(by[-1]) != 0xFF, i
        zeros = int(ff.sum())
        assert k == int(res[i]["total_bits_unstuffed"]) // 8 + zeros, i
    for dst, src in twins.items():
        assert res[dst].tobytes() == res[src].tobytes()
        k = int(res[src]["n_full_words"]) + 1
        assert torch.equal(d_words[dst, :k], d_words[src, :k])
    checked = (0, 77, 128, 254)
    host = {i: frames[i].cpu().numpy() for i in checked}
    got = {i: d_words[i, :int(res[i]["n_full_words"]) + (1 if res[i]["tail_bits"] else 0)].cpu().numpy().view(np.uint32)
           for i in checked}
    with ThreadPoolExecutor(4) as pool:
        list(pool.map(lambda i: same_as_oracle(got[i], res[i], host[i]), checked))
