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
    """encode_stripe(s) -> (words tensor, result mapping) for stripe s; encodes this rank's stripes and
    gathers the container on root."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    local = {s: encode_stripe(s) for s in stripes_of_rank(n_stripes, rank, world)}
    return gather_stripes(local, n_stripes, device=device, root=root, group=group)


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
