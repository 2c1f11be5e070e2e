"""The entry point that allocates nothing (jpegenc_encode_batch_noalloc, static workspace inside the
library) against jpegenc_encode_batch on the same inputs: word for word, through the C ABI, twice in a
row with the second input smaller (the static array still holds the first call's contents); and the same entry
point captured into a HIP graph (capture refuses allocation, so that also shows the call makes none) and
replayed on new pixels in the same buffers."""
import numpy as np
import pytest

import oracle_lib as O
from conftest import load_pkg

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def run_both(pkg, e, frames, cap):
    """Codes `frames` (n, h, w, 3 uint8 on the GPU) through both entry points; returns [(words, results)] * 2."""
    n, h, w, _ = frames.shape
    outs = []
    for call in (e.encode_batch_device, e.encode_batch_noalloc):
        d_words = torch.full((n, cap), 0x5A5A5A5A, dtype=torch.int32, device=frames.device)
        d_res = torch.zeros((n, 32), dtype=torch.uint8, device=frames.device)
        torch.cuda.synchronize()
        call(frames.data_ptr(), n, w, h, w * 3, h * w * 3, d_words.data_ptr(), cap, d_res.data_ptr())
        torch.cuda.synchronize()
        outs.append((d_words.cpu().numpy().view(np.uint32), d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(n)))
    return outs


def test_noalloc_entry_point_matches_the_allocating_one():
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    big = pkg.synth.natural_like_torch(3, 1080, 1920, 500, dev)           # 3 x 507 chunks
    big[1, 300:500, 400:900] = torch.randint(0, 256, (200, 500, 3), generator=g, device=dev, dtype=torch.uint8)
    small = torch.randint(0, 256, (2, 72, 136, 3), generator=g, device=dev, dtype=torch.uint8)  # noise: long strings
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        for frames in (big, small):  # two calls in a row, the second smaller
            n, h, w, _ = frames.shape
            cap = pkg.max_words(w, h) // 2
            (w_ref, r_ref), (w_new, r_new) = run_both(pkg, e, frames, cap)
            assert (r_ref["status"] == 0).all()
            assert r_new.tobytes() == r_ref.tobytes()
            assert np.array_equal(w_new, w_ref)  # every word of the output buffers, written or not
            # and the allocating entry point itself is right: one frame of each against the oracle
            k = int(r_ref[n - 1]["n_full_words"]) + (1 if r_ref[n - 1]["tail_bits"] else 0)
            want, r = O.encode(frames[n - 1].cpu().numpy())
            assert np.array_equal(w_new[n - 1, :k], want) and int(r_new[n - 1]["tail_word"]) == r.tail_word
        # too small an output buffer is reported per image, as by the allocating entry point
        n, h, w, _ = small.shape
        (w_ref, r_ref), (w_new, r_new) = run_both(pkg, e, small, 64)
        assert (r_ref["status"] == pkg.E_TOO_SMALL).all() and r_new.tobytes() == r_ref.tobytes()
        assert np.array_equal(w_new, w_ref)
        # sizes beyond the static workspace are refused, nothing is enqueued
        five = pkg.synth.natural_like_torch(5, 1080, 1920, 600, dev)  # 5 x 507 = 2535 > 2048 chunks
        d_words = torch.zeros((5, 16), dtype=torch.int32, device=dev)
        d_res = torch.zeros((5, 32), dtype=torch.uint8, device=dev)
        with pytest.raises(pkg.JpegEncError) as err:
            e.encode_batch_noalloc(five.data_ptr(), 5, 1920, 1080, 1920 * 3, 1080 * 1920 * 3, d_words.data_ptr(), 16,
                                   d_res.data_ptr())
        assert err.value.code == pkg.E_LIMIT
        torch.cuda.synchronize()
        assert not d_words.any() and not d_res.any()
        # exactly at the limit: one 3840 x 2160 frame is 2025 chunks
        uhd = pkg.synth.natural_like_torch(1, 2160, 3840, 601, dev)
        (w_ref, r_ref), (w_new, r_new) = run_both(pkg, e, uhd, 3840 * 2160 * 3 // 4)
        assert (r_ref["status"] == 0).all() and r_new.tobytes() == r_ref.tobytes() and np.array_equal(w_new, w_ref)
    # a second context shares the device's array; used one after the other that is fine
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream, qtabs=[[4] * 64] * 3) as e2:
        (w_ref, r_ref), (w_new, r_new) = run_both(pkg, e2, small, pkg.max_words(136, 72))
        assert (r_ref["status"] == 0).all() and r_new.tobytes() == r_ref.tobytes() and np.array_equal(w_new, w_ref)


@pytest.mark.parametrize("mode", ["auto", "small", "large", "short", "quant"])
def test_noalloc_call_replays_from_a_captured_graph(mode):
    """The call captured once on a side stream, then replayed three times on other pixels in the same buffers:
    every replay gives the oracle's words (nothing in the launches is tied to the call that was captured).  With
    the staging-region size chosen by the context or forced, and with quantisation tables; afterwards the context
    moves to another stream and codes as ever (the capture has left no event behind for it to wait on)."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    n, h, w = 3, 264, 520  # 3 x 34 chunks, the last one of an image partial
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    contents = [pkg.synth.natural_like_torch(n, h, w, 900, dev),
                torch.randint(0, 256, (n, h, w, 3), generator=gen, device=dev, dtype=torch.uint8),
                pkg.synth.natural_like_torch(n, h, w, 901, dev)]
    cap = pkg.max_words(w, h)
    buf = contents[0].clone()
    d_words = torch.zeros((n, cap), dtype=torch.int32, device=dev)
    d_res = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    args = (buf.data_ptr(), n, w, h, w * 3, h * w * 3, d_words.data_ptr(), cap, d_res.data_ptr())
    qtabs = [[3 + (i % 7) for i in range(64)]] * 3 if mode == "quant" else None
    cfg = O.default_config(qtabs=qtabs)
    with pkg.JpegTop(stream=side.cuda_stream, qtabs=qtabs) as e:
        if mode in ("small", "large", "short"):
            e.set_region_mode({"small": e.REGIONS_SMALL, "large": e.REGIONS_LARGE, "short": e.REGIONS_SHORT}[mode])
        e.encode_batch_noalloc(*args)  # once outside capture: code objects loaded, nothing left to set up
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            e.encode_batch_noalloc(*args)
        for frames in contents[1:] + contents[:1]:
            buf.copy_(frames)
            d_words.fill_(0x5A5A5A5A)
            d_res.zero_()
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(n)
            assert (res["status"] == 0).all()
            host = frames.cpu().numpy()
            for i in range(n):
                want, r = O.encode(host[i], cfg)
                k = int(res[i]["n_full_words"]) + (1 if res[i]["tail_bits"] else 0)
                assert k == len(want) and np.array_equal(d_words[i, :k].cpu().numpy().view(np.uint32), want), i
                assert int(res[i]["tail_word"]) == r.tail_word
        # on to another stream and an ordinary call
        other = torch.cuda.Stream()
        e.set_stream(other.cuda_stream)
        d_words.fill_(0)
        torch.cuda.synchronize()
        e.encode_batch_device(*args)
        torch.cuda.synchronize()
        res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(n)
        want, r = O.encode(buf[0].cpu().numpy(), cfg)
        assert np.array_equal(d_words[0, :len(want)].cpu().numpy().view(np.uint32), want) and int(res[0]["status"]) == 0
        del graph


def test_reserved_calls_allocate_nothing():
    """After jpegenc_reserve a device-pointer call of at most that size leaves the device's free memory where it
    was -- the one-frame call included, which runs as a single launch over three times the chunks."""
    pkg = load_pkg()
    dev = torch.device("cuda:0")
    frames = pkg.synth.natural_like_torch(2, 1080, 1920, 31, dev)
    cap = pkg.max_words(1920, 1080) // 2
    d_words = torch.zeros((2, cap), dtype=torch.int32, device=dev)
    d_res = torch.zeros((2, 32), dtype=torch.uint8, device=dev)
    with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
        e.reserve(2, 1920, 1080)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for n in (1, 2, 1):
            e.encode_batch_device(frames.data_ptr(), n, 1920, 1080, 1920 * 3, 1080 * 1920 * 3, d_words.data_ptr(), cap,
                                  d_res.data_ptr())
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] == free0
        res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(2)
        want, r = O.encode(frames[0].cpu().numpy())
        assert np.array_equal(d_words[0, :len(want)].cpu().numpy().view(np.uint32), want) and int(res[0]["status"]) == 0
