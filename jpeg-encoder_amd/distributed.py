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
