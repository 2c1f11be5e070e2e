"""Multi-GPU partitioning of the jpeg_top path: one process per GPU, torch.distributed (RCCL over xGMI
on the GPUs, gloo in CPU tests).

The reference has no notion of parallelism (one clock domain, one module tree); what makes sharding
legal is its reset semantics: every image starts from `rst` (DC predictors and bit offset zero,
rtl/4_huffman/cr_huff.sv:706-717), so images are independent units, and a horizontal stripe of a huge
image encoded after `rst` is exactly a restart interval.

* batch of images  -> contiguous image ranges per rank, no data-path collective (`shard_range`).
* one huge image   -> stripe s goes to rank s % world; the only exchange step is the reassembly of the
  variable-length stripe streams on the root: one all_gather of the stripe tables (a few words per
  stripe) and one grouped send/recv of the payloads (`gather_stripes`).
* one huge image, one stream -> band r (whole block rows) goes to rank r; the bands are coupled by what
  the RTL carries from block to block -- DC predictors (cr_huff.sv:591-596: read from the 8 pixel rows
  above the band), running bit offset (fifo_out.sv:222-243) and stuffed byte count
  (ff_checker.sv:96-179) -- so two tiny all_gathers (bit counts, then 0xFF counts) sit between the three
  phases of the band coder, and the root ORs the bands' words together (`encode_monolithic`).
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous range [lo, hi) of rank `rank` when n_items are split over `world` ranks."""
    lo = (n_items * rank) // world
    hi = (n_items * (rank + 1)) // world
    return lo, hi


def stripes_of_rank(n_stripes, rank, world):
    return list(range(rank, n_stripes, world))


STRIPE_FIELDS = ("n_full_words", "tail_bits", "tail_word", "total_bits_unstuffed", "total_bytes_stuffed", "n_words")


def gather_stripes(local, n_stripes, device=None, root=0, group=None):
    """Reassembles a striped image on `root`.

    local: dict stripe_index -> (words, result) for the stripes this rank encoded, where `words` is a
    1-D torch int32/uint32-bit-pattern tensor (on `device`) holding the full words followed by the tail
    word, and `result` is a mapping with the jpegenc_result fields.
    Returns on root: (table, payload) -- table is an (n_stripes, 7) int64 array with columns
    word_offset + STRIPE_FIELDS, payload the concatenated words of all stripes in stripe order (torch
    tensor on `device`); on the other ranks (None, None).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    # 1. stripe tables: every rank contributes its rows, all_gather makes the full table everywhere
    mine = torch.zeros((n_stripes, len(STRIPE_FIELDS)), dtype=torch.int64, device=dev)
    for s, (words, res) in local.items():
        row = [int(res["n_full_words"]), int(res["tail_bits"]), int(res["tail_word"]),
               int(res["total_bits_unstuffed"]), int(res["total_bytes_stuffed"]), int(words.numel())]
        mine[s] = torch.tensor(row, dtype=torch.int64, device=dev)
    tables = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(tables, mine, group=group)
    table = torch.stack(tables).sum(0)  # every stripe is owned by exactly one rank
    n_words = table[:, STRIPE_FIELDS.index("n_words")].cpu().numpy()
    offsets = np.concatenate([[0], np.cumsum(n_words)]).astype(np.int64)
    # 2. payloads: grouped point-to-point into the root's buffer (7 senders use 7 distinct xGMI links)
    ops = []
    payload = None
    if rank == root:
        payload = torch.zeros(int(offsets[-1]), dtype=torch.int32, device=dev)
        for s in range(n_stripes):
            owner = s % world
            view = payload[int(offsets[s]):int(offsets[s + 1])]
            if owner == root:
                view.copy_(local[s][0].view(torch.int32))
            elif view.numel():
                ops.append(dist.P2POp(dist.irecv, view, owner, group=group, tag=s))
    else:
        for s, (words, _) in sorted(local.items()):
            if words.numel():
                ops.append(dist.P2POp(dist.isend, words.view(torch.int32).contiguous(), root, group=group, tag=s))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != root:
        return None, None
    out = np.concatenate([offsets[:-1, None], table.cpu().numpy()], axis=1)
    return out, payload


def encode_striped_image(encode_stripe, n_stripes, device=None, root=0, group=None):
    """Encodes this rank's stripes and gathers the container on root.  `encode_stripe` is either a function
    encode_stripe(s) -> (words tensor, result mapping) for stripe s, or a coder with
    encode_all() -> (words, results): all of the rank's stripes in one go, stripe stripes_of_rank(...)[k] in row k
    of the (n_local, cap) int32 tensor `words`, its 32-byte result record in row k of the (n_local, 32) uint8
    tensor `results`, both still on `device` -- then the records reach the host once, after they have been
    gathered (`gather_stripes_resident`)."""
    import torch.distributed as dist
    if hasattr(encode_stripe, "encode_all"):
        words, results = encode_stripe.encode_all()
        return gather_stripes_resident(words, results, n_stripes, device=device, root=root, group=group)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    local = {s: encode_stripe(s) for s in stripes_of_rank(n_stripes, rank, world)}
    return gather_stripes(local, n_stripes, device=device, root=root, group=group)


def gather_stripes_resident(words, results, n_stripes, device=None, root=0, group=None):
    """gather_stripes for stripes whose words and result records are still where they were coded (see
    encode_striped_image): one all_gather of the raw records, one copy of the gathered table to the host -- the
    only point where the host waits for the coder -- then the grouped send/recv of the payloads.  Same return
    value as gather_stripes."""
    import torch
    import torch.distributed as dist
    from .encoder import RESULT_DTYPE
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    mine = stripes_of_rank(n_stripes, rank, world)
    per = (n_stripes + world - 1) // world
    rec = torch.zeros((per, 32), dtype=torch.uint8, device=dev)
    rec[:len(mine)] = results.reshape(len(mine), 32)
    if world > 1:
        parts = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(parts, rec, group=group)
        every = torch.stack(parts)
    else:
        every = rec[None]
    every = every.cpu().numpy().view(RESULT_DTYPE).reshape(world, per)  # the host waits here, once
    recs = [every[s % world, s // world] for s in range(n_stripes)]
    bad = [s for s in range(n_stripes) if int(recs[s]["status"]) != 0]
    if bad:
        raise RuntimeError("stripes %s do not fit their %d words" % (bad, words.shape[1]))
    n_words = np.array([int(r["n_full_words"]) + (1 if r["tail_bits"] else 0) for r in recs], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(n_words)]).astype(np.int64)
    ops, payload = [], None
    if rank == root:
        payload = torch.empty(int(offsets[-1]), dtype=torch.int32, device=dev)
        for s in range(n_stripes):
            view = payload[int(offsets[s]):int(offsets[s + 1])]
            if s % world == root:
                view.copy_(words[s // world, :int(n_words[s])].view(torch.int32))
            elif view.numel():
                ops.append(dist.P2POp(dist.irecv, view, s % world, group=group, tag=s))
    else:
        for k, s in enumerate(mine):
            if n_words[s]:
                ops.append(dist.P2POp(dist.isend, words[k, :int(n_words[s])].view(torch.int32), root, group=group, tag=s))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != root:
        return None, None
    table = np.array([[int(offsets[s])] + [int(recs[s][f]) for f in STRIPE_FIELDS[:-1]] + [int(n_words[s])]
                      for s in range(n_stripes)], dtype=np.int64)
    return table, payload


def band_rows(h, rank, world):
    """Rows [lo, hi) of band `rank` when an image of h rows (a multiple of 8) is cut into `world` bands of
    whole block rows."""
    lo, hi = shard_range(h // 8, rank, world)
    return 8 * lo, 8 * hi


def encode_monolithic(band, device=None, root=0, group=None):
    """One stream for one image from one band per rank (band r directly above band r + 1).

    `band` is this rank's band coder, an object with the three phases of the C ABI's segment calls:
        band.encode() -> (bits, tail8)
        band.count(bit_offset, prev_tail8) -> ff_bytes
        band.write(ff_offset, is_last) -> (first_word, words, result or None)
    with `words` a 1-D uint32 numpy array (or torch tensor of int32 bit patterns on `device`).
    Returns on root (words, result): the words of jpeg_top's single run over the whole image and the
    result record of the last band (a mapping/record with the jpegenc_result fields); elsewhere
    (None, None)."""
    import torch
    import torch.distributed as dist
    if getattr(band, "device_resident", False):
        return encode_monolithic_resident(band, device=device, root=root, group=group)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")

    def gather_row(values):
        mine = torch.tensor(values, dtype=torch.int64, device=dev)
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine, group=group)
        return torch.stack(rows).cpu().numpy()

    bits, tail8 = band.encode()
    t = gather_row([bits, tail8])
    bit_offset = int(t[:rank, 0].sum())
    prev_tail = int(t[rank - 1, 1]) if rank else 0
    ff = band.count(bit_offset, prev_tail)
    f = gather_row([ff])
    first_word, words, result = band.write(int(f[:rank, 0].sum()), rank == world - 1)
    if not torch.is_tensor(words):
        words = torch.from_numpy(np.ascontiguousarray(words).view(np.int32))
    words = words.view(torch.int32).to(dev)
    res_row = [0] * 7
    if result is not None:
        res_row = [int(result[k]) for k in ("n_full_words", "total_bits_unstuffed", "total_bytes_stuffed", "tail_word",
                                            "tail_bits", "eof_partial_ready", "status")]
    spans = gather_row([first_word, int(words.numel())] + res_row)
    total = int((spans[:, 0] + spans[:, 1]).max())
    ops, parts = [], {}
    if rank == root:
        for r in range(world):
            if r == root:
                parts[r] = words
            else:
                parts[r] = torch.zeros(int(spans[r, 1]), dtype=torch.int32, device=dev)
                if parts[r].numel():
                    ops.append(dist.P2POp(dist.irecv, parts[r], r, group=group, tag=r))
    elif words.numel():
        ops.append(dist.P2POp(dist.isend, words.contiguous(), root, group=group, tag=rank))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != root:
        return None, None
    out = torch.zeros(total, dtype=torch.int32, device=dev)
    for r in range(world):  # neighbouring bands share their boundary words: each wrote only its own bytes
        a = int(spans[r, 0])
        out[a:a + parts[r].numel()] |= parts[r]
    last = spans[world - 1]
    if int(last[6]):  # the tail word may be shared with the band before the last: take it from the merged words
        last[5] = int(out[int(last[2])].item()) & 0xFFFFFFFF
    result = dict(n_full_words=int(last[2]), total_bits_unstuffed=int(last[3]), total_bytes_stuffed=int(last[4]),
                  tail_word=int(last[5]), tail_bits=int(last[6]), eof_partial_ready=int(last[7]), status=int(last[8]))
    return out, result


def encode_monolithic_resident(band, device=None, root=0, group=None):
    """encode_monolithic for a band coder that keeps its figures where it computes them (`band.device_resident`):
        band.encode() -> int64 tensor [bits, tail8] on `device`
        band.count(bit_offset, prev_tail8) -> int64 tensor [ff_bytes]      (arguments: one-element int64 tensors)
        band.write(ff_offset, is_last) -> (int64 tensor [first_word, n_words], uint8 tensor of the 32-byte result
                                           record, int32 tensor of words with room to spare)
    The offsets are sums over the gathered tensors, taken on `device`; the host sees nothing until the bands'
    spans have been gathered and it has to size the receives -- one wait per image instead of one per phase."""
    import torch
    import torch.distributed as dist
    from .encoder import RESULT_DTYPE
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")

    def gather(t):
        if world == 1:
            return t[None]
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous(), group=group)
        return torch.stack(parts)

    rows = gather(band.encode())                                  # [world, 2]
    bit_offset = rows[:rank, 0].sum().reshape(1)
    prev_tail = rows[rank - 1, 1].reshape(1) if rank else torch.zeros(1, dtype=torch.int64, device=dev)
    ff = gather(band.count(bit_offset, prev_tail))                # [world, 1]
    span, rec, words = band.write(ff[:rank, 0].sum().reshape(1), rank == world - 1)
    room = torch.tensor([words.numel()], dtype=torch.int64).to(dev)
    packed = gather(torch.cat([span.view(torch.uint8), room.view(torch.uint8), rec.view(torch.uint8)]))  # [world, 24 + 32]
    packed = packed.cpu().numpy()                                 # the host waits here, once
    spans = packed[:, :24].copy().view(np.int64).reshape(world, 3)
    last = packed[world - 1, 24:].copy().view(RESULT_DTYPE)[0]
    if int((spans[:, 1] > spans[:, 2]).any()):  # (every rank sees every band's figures: all raise or none)
        raise RuntimeError("bands %s do not fit their words" % np.nonzero(spans[:, 1] > spans[:, 2])[0].tolist())
    words = words[:int(spans[rank, 1])].view(torch.int32)
    total = int((spans[:, 0] + spans[:, 1]).max())
    ops, parts = [], {}
    if rank == root:
        for r in range(world):
            if r == root:
                parts[r] = words
            else:
                parts[r] = torch.empty(int(spans[r, 1]), dtype=torch.int32, device=dev)
                if parts[r].numel():
                    ops.append(dist.P2POp(dist.irecv, parts[r], r, group=group, tag=r))
    elif words.numel():
        ops.append(dist.P2POp(dist.isend, words.contiguous(), root, group=group, tag=rank))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != root:
        return None, None
    if world == 1:
        out = words  # one band: its words are the stream
    else:
        out = torch.zeros(total, dtype=torch.int32, device=dev)
        for r in range(world):  # neighbouring bands share their boundary words: each wrote only its own bytes
            a = int(spans[r, 0])
            out[a:a + parts[r].numel()] |= parts[r]
    result = {k: int(last[k]) for k in ("n_full_words", "total_bits_unstuffed", "total_bytes_stuffed", "tail_word",
                                        "tail_bits", "eof_partial_ready", "status")}
    if result["tail_bits"]:  # the tail word may be shared with the band before the last: take it from the merged words
        result["tail_word"] = int(out[result["n_full_words"]].item()) & 0xFFFFFFFF
    return out, result


# ---- the HIP path behind the functions above: device-resident images, torch tensors in and out ------------
def hip_stripe_coder(enc, image, n_stripes, bytes_per_pixel_room=3.0):
    """Returns encode_stripe(s) for `encode_striped_image`: stripe s of `image` (an (h, w, 3) uint8 torch
    tensor on the encoder's GPU; h a multiple of 8 * n_stripes) coded by `enc` (a JpegTop whose stream is
    torch's current one) as jpeg_top would code that sub-image after rst.  The words stay on the GPU."""
    import torch
    from .encoder import RESULT_DTYPE
    h, w, _ = image.shape
    rows = h // n_stripes
    assert rows * n_stripes == h and rows % 8 == 0 and image.is_contiguous() and image.dtype == torch.uint8
    cap = int(w * rows * bytes_per_pixel_room) // 4 + 2

    def encode_stripe(s):
        words = torch.zeros(cap, dtype=torch.int32, device=image.device)
        res_d = torch.zeros(32, dtype=torch.uint8, device=image.device)
        enc.encode_batch_device(image.data_ptr() + s * rows * w * 3, 1, w, rows, w * 3, rows * w * 3,
                                words.data_ptr(), cap, res_d.data_ptr())
        enc.synchronize()
        res = res_d.cpu().numpy().view(RESULT_DTYPE)[0]
        if int(res["status"]) != 0:
            raise RuntimeError("stripe %d does not fit %d words" % (s, cap))
        n = int(res["n_full_words"]) + (1 if res["tail_bits"] else 0)
        return words[:n], res

    return encode_stripe


class HipStripes:
    """Coder for `encode_striped_image` on the HIP path: this rank's stripes, one under the other in `local_img`
    (an (n_local * rows, w, 3) uint8 torch tensor on the encoder's GPU), coded by ONE jpegenc_encode_batch call
    (every stripe an image of its own: as jpeg_top codes that sub-image after rst); words and result records stay
    on the GPU.  `enc`: a JpegTop whose stream is torch's current one."""

    def __init__(self, enc, local_img, n_local, bytes_per_pixel_room=3.0):
        import torch
        h, w, _ = local_img.shape
        self.rows = h // n_local
        assert self.rows * n_local == h and self.rows % 8 == 0 and local_img.is_contiguous() and local_img.dtype == torch.uint8
        self.enc, self.img, self.n, self.w = enc, local_img, n_local, w
        self.cap = int(w * self.rows * bytes_per_pixel_room) // 4 + 2
        enc.reserve(n_local, w, self.rows)

    def encode_all(self):
        import torch
        words = torch.empty((self.n, self.cap), dtype=torch.int32, device=self.img.device)
        res = torch.empty((self.n, 32), dtype=torch.uint8, device=self.img.device)
        self.enc.encode_batch_device(self.img.data_ptr(), self.n, self.w, self.rows, self.w * 3, self.rows * self.w * 3,
                                     words.data_ptr(), self.cap, res.data_ptr())
        return words, res


class HipBandResident:
    """Band coder for `encode_monolithic` on the HIP path that never waits for the GPU: rows [lo, hi) of `image`
    through the three device-resident segment calls of the C ABI; bit counts, offsets, spans and words are torch
    tensors on the encoder's GPU.  `enc`: a JpegTop whose stream is torch's current one, so that the library's
    launches and torch's small tensor operations run in the order they are issued."""
    device_resident = True

    def __init__(self, enc, image, lo, hi, bytes_per_pixel_room=3.0):
        import torch
        assert image.is_contiguous() and image.dtype == torch.uint8 and lo % 8 == 0 and hi % 8 == 0
        self.enc, self.image, self.lo, self.hi = enc, image, lo, hi
        self.w = image.shape[1]
        self.cap = int(self.w * (hi - lo) * bytes_per_pixel_room) // 4 + 4
        # bits, tail8 | bit_offset, prev_tail8 | ff_bytes | ff_offset | first_word, n_words
        self.fig = torch.zeros(8, dtype=torch.int64, device=image.device)
        self.rec = torch.zeros(32, dtype=torch.uint8, device=image.device)

    def _at(self, i):
        return self.fig.data_ptr() + 8 * i

    def encode(self):
        self.enc.segment_encode_dev(self.image.data_ptr() + self.lo * self.w * 3, self.w, self.hi - self.lo, self.w * 3,
                                    self.lo > 0, self._at(0))
        return self.fig[0:2]

    def count(self, bit_offset, prev_tail8):
        self.fig[2:3].copy_(bit_offset)
        self.fig[3:4].copy_(prev_tail8)
        self.enc.segment_count_dev(self._at(2), self._at(3), self._at(4))
        return self.fig[4:5]

    def write(self, ff_offset, is_last):
        import torch
        self.fig[5:6].copy_(ff_offset)
        words = torch.empty(self.cap, dtype=torch.int32, device=self.image.device)
        self.enc.segment_write_dev(self._at(5), is_last, words.data_ptr(), self.cap, self._at(6), self.rec.data_ptr())
        return self.fig[6:8], self.rec, words


class HipBand:
    """Band coder for `encode_monolithic` on the HIP path: rows [lo, hi) of `image` (an (h, w, 3) uint8 torch
    tensor on the encoder's GPU) through the three segment calls of the C ABI; the words stay on the GPU."""

    def __init__(self, enc, image, lo, hi, bytes_per_pixel_room=3.0):
        import torch
        assert image.is_contiguous() and image.dtype == torch.uint8 and lo % 8 == 0 and hi % 8 == 0
        self.enc, self.image, self.lo, self.hi = enc, image, lo, hi
        self.w = image.shape[1]
        self.cap = int(self.w * (hi - lo) * bytes_per_pixel_room) // 4 + 4

    def encode(self):
        return self.enc.segment_encode_device(self.image.data_ptr() + self.lo * self.w * 3, self.w, self.hi - self.lo,
                                              self.w * 3, self.lo > 0)

    def count(self, bit_offset, prev_tail8):
        return self.enc.segment_count(bit_offset, prev_tail8)

    def write(self, ff_offset, is_last):
        import torch
        words = torch.zeros(self.cap, dtype=torch.int32, device=self.image.device)
        torch.cuda.current_stream(self.image.device).synchronize()
        first_word, n_words, result = self.enc.segment_write_device(ff_offset, is_last, words.data_ptr(), self.cap)
        return first_word, words[:n_words], result
