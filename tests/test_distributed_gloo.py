"""The N > 1 paths on CPU: two gloo ranks.  The encoder behind the sharding logic is the CPU oracle here
(test infrastructure); on the GPUs the same functions are fed by the HIP path."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from conftest import ROOT, load_pkg, natural_like  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    pkg = load_pkg()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        if mode == "stripes":
            img = natural_like(96, 72, 5)
            n_stripes, rows = 4, 24

            def enc(s):
                words, r = O.encode(img[rows * s:rows * (s + 1)])
                res = dict(n_full_words=r.n_full_words, tail_bits=r.tail_bits, tail_word=r.tail_word,
                           total_bits_unstuffed=r.total_bits_unstuffed, total_bytes_stuffed=r.total_bytes_stuffed)
                return torch.from_numpy(words.view(np.int32).copy()), res

            table, payload = pkg.distributed.encode_striped_image(enc, n_stripes)
            if rank == 0:
                np.save(os.path.join(out_dir, "table.npy"), table)
                np.save(os.path.join(out_dir, "payload.npy"), payload.numpy().view(np.uint32))
            else:
                assert table is None and payload is None
        elif mode == "stripes_resident":
            # the coder hands over all of the rank's stripes at once, records unread: what the HIP path does
            img = natural_like(96, 72, 5)
            n_stripes, rows = 4, 24

            class Coder:
                def encode_all(self):
                    mine = pkg.distributed.stripes_of_rank(n_stripes, rank, world)
                    words = torch.full((len(mine), 4000), 0x5A5A5A5A, dtype=torch.int32)
                    recs = np.zeros(len(mine), dtype=pkg.RESULT_DTYPE)
                    for k, s in enumerate(mine):
                        w, r = O.encode(img[rows * s:rows * (s + 1)])
                        words[k, :len(w)] = torch.from_numpy(w.view(np.int32).copy())
                        for f in ("n_full_words", "tail_bits", "tail_word", "total_bits_unstuffed", "total_bytes_stuffed"):
                            recs[k][f] = getattr(r, f)
                    return words, torch.from_numpy(recs.view(np.uint8).reshape(len(mine), 32).copy())

            table, payload = pkg.distributed.encode_striped_image(Coder(), n_stripes)
            if rank == 0:
                np.save(os.path.join(out_dir, "table.npy"), table)
                np.save(os.path.join(out_dir, "payload.npy"), payload.numpy().view(np.uint32))
            else:
                assert table is None and payload is None
        elif mode == "mono_resident":
            for k, img in enumerate(mono_images()):
                lo, hi = pkg.distributed.band_rows(img.shape[0], rank, world)
                words, res = pkg.distributed.encode_monolithic(O.OracleBandResident(img, lo, hi))
                if rank == 0:
                    np.save(os.path.join(out_dir, "mono%d.npy" % k), words.numpy().view(np.uint32))
                    np.save(os.path.join(out_dir, "mono%d_res.npy" % k),
                            np.array([res["n_full_words"], res["tail_bits"], res["tail_word"],
                                      res["total_bits_unstuffed"], res["total_bytes_stuffed"]], dtype=np.int64))
                else:
                    assert words is None and res is None
        elif mode == "mono":
            # one stream from one band per rank; the band coder is the oracle stand-in
            for k, img in enumerate(mono_images()):
                lo, hi = pkg.distributed.band_rows(img.shape[0], rank, world)
                words, res = pkg.distributed.encode_monolithic(O.OracleBand(img, lo, hi))
                if rank == 0:
                    np.save(os.path.join(out_dir, "mono%d.npy" % k), words.numpy().view(np.uint32))
                    np.save(os.path.join(out_dir, "mono%d_res.npy" % k),
                            np.array([res["n_full_words"], res["tail_bits"], res["tail_word"],
                                      res["total_bits_unstuffed"], res["total_bytes_stuffed"]], dtype=np.int64))
                else:
                    assert words is None and res is None
        else:
            # batch sharding: every rank encodes its contiguous range; rank 0 checks the ranges tile the batch
            n = 7
            lo, hi = pkg.distributed.shard_range(n, rank, world)
            sizes = torch.zeros(n, dtype=torch.int64)
            for i in range(lo, hi):
                img = np.random.default_rng(50 + i).integers(0, 256, (16, 24, 3), dtype=np.uint8)
                words, r = O.encode(img)
                sizes[i] = len(words)
            dist.all_reduce(sizes)  # optional packed-index exchange: one tiny collective
            if rank == 0:
                np.save(os.path.join(out_dir, "sizes.npy"), sizes.numpy())
    finally:
        dist.destroy_process_group()


def mono_images():
    return [natural_like(96, 72, 5), np.random.default_rng(8).integers(0, 256, (80, 64, 3), dtype=np.uint8),
            np.full((16, 8, 3), 255, np.uint8)]


def run(mode, tmp_path):
    port = free_port()
    mp.spawn(worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)


def test_shard_ranges_tile_the_batch():
    pkg = load_pkg()
    for n in (1, 7, 8, 8192, 1000):
        for world in (1, 2, 4, 8):
            spans = [pkg.distributed.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert pkg.distributed.shard_range(8192, 3, 8) == (3072, 4096)
    assert pkg.distributed.stripes_of_rank(8, 1, 2) == [1, 3, 5, 7]


@pytest.mark.parametrize("mode", ["stripes", "stripes_resident"])
def test_stripe_gather_two_ranks(tmp_path, mode):
    import oracle_lib as O
    run(mode, tmp_path)
    table = np.load(tmp_path / "table.npy")
    payload = np.load(tmp_path / "payload.npy")
    img = natural_like(96, 72, 5)
    off = 0
    for s in range(4):
        words, r = O.encode(img[24 * s:24 * (s + 1)])
        row = table[s]
        assert row[0] == off
        assert (row[1], row[2], row[3]) == (r.n_full_words, r.tail_bits, r.tail_word)
        assert row[6] == len(words)
        assert np.array_equal(payload[off:off + len(words)], words)
        off += len(words)
    assert off == len(payload)


def test_batch_sharding_two_ranks(tmp_path):
    import oracle_lib as O
    run("batch", tmp_path)
    sizes = np.load(tmp_path / "sizes.npy")
    for i in range(7):
        img = np.random.default_rng(50 + i).integers(0, 256, (16, 24, 3), dtype=np.uint8)
        assert sizes[i] == len(O.encode(img)[0])


@pytest.mark.parametrize("mode", ["mono", "mono_resident"])
def test_monolithic_stream_from_two_bands(tmp_path, mode):
    """Two ranks, one band each: the ORed words are the whole image's single stream.  With the figures passed as
    integers through the host, and kept in tensors until the spans have been gathered."""
    import oracle_lib as O
    run(mode, tmp_path)
    for k, img in enumerate(mono_images()):
        want, r = O.encode(img)
        got = np.load(tmp_path / ("mono%d.npy" % k))
        res = np.load(tmp_path / ("mono%d_res.npy" % k))
        assert np.array_equal(got, want)
        assert tuple(res) == (r.n_full_words, r.tail_bits, r.tail_word, r.total_bits_unstuffed, r.total_bytes_stuffed)


def test_oracle_band_stand_in_any_cut():
    """The stand-in itself, without ranks: every cut of a small image into bands reproduces the stream."""
    import oracle_lib as O
    img = np.random.default_rng(12).integers(0, 256, (40, 24, 3), dtype=np.uint8)
    img[8:16] = 255  # long runs of one-bits around a cut
    want, r = O.encode(img)
    for cuts in ([0, 40], [0, 8, 40], [0, 16, 24, 40], [0, 8, 16, 24, 32, 40]):
        bands = [O.OracleBand(img, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
        info = [b.encode() for b in bands]
        offs = np.concatenate([[0], np.cumsum([i[0] for i in info])])
        ffs = [b.count(int(offs[i]), info[i - 1][1] if i else 0) for i, b in enumerate(bands)]
        ffo = np.concatenate([[0], np.cumsum(ffs)])
        out = np.zeros(len(want) + 2, dtype=np.uint32)
        for i, b in enumerate(bands):
            fw, words, res = b.write(int(ffo[i]), i == len(bands) - 1)
            out[fw:fw + len(words)] |= words
        assert np.array_equal(out[:len(want)], want)
        assert (res["n_full_words"], res["tail_bits"]) == (r.n_full_words, r.tail_bits)
