#!/usr/bin/env python3
"""One-off check of the largest configuration: a 16384 x 16384 image coded as one stream on the GPU, against
the CPU oracle and against the same image coded as 8 bands merged (run on a GPU box)."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

pkg = importlib.import_module("jpeg-encoder_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda:0")
img_d = torch.cat([pkg.synth.natural_like_torch(1, n // 8, n, 7 + k, dev)[0] for k in range(8)])  # (n, n, 3)
torch.cuda.synchronize()
print("image", tuple(img_d.shape), flush=True)
cap = pkg.max_words(n, n) // 3
words_d = torch.zeros(cap, dtype=torch.int32, device=dev)
res_d = torch.zeros(32, dtype=torch.uint8, device=dev)
with pkg.JpegTop(stream=torch.cuda.current_stream().cuda_stream) as e:
    t0 = time.time()
    e.encode_batch_device(img_d.data_ptr(), 1, n, n, n * 3, n * n * 3, words_d.data_ptr(), cap, res_d.data_ptr())
    torch.cuda.synchronize()
    print("gpu whole image: %.1f ms" % ((time.time() - t0) * 1e3), flush=True)
res = res_d.cpu().numpy().view(pkg.RESULT_DTYPE)[0]
k = int(res["n_full_words"]) + (1 if res["tail_bits"] else 0)
whole = words_d[:k].cpu().numpy().view(np.uint32)
print("status", int(res["status"]), "words", k, "B/px %.3f" % (int(res["total_bytes_stuffed"]) / n / n), flush=True)

# 8 bands, one context each, device pointers
cuts = [n // 8 * i for i in range(9)]
bands = [pkg.JpegTop() for _ in range(8)]
info = [b.segment_encode_device(img_d.data_ptr() + a * n * 3, n, c - a, n * 3, a > 0) for b, a, c in zip(bands, cuts[:-1], cuts[1:])]
offs = np.concatenate([[0], np.cumsum([i[0] for i in info])])
ffs = [b.segment_count(int(offs[i]), info[i - 1][1] if i else 0) for i, b in enumerate(bands)]
ffo = np.concatenate([[0], np.cumsum(ffs)])
merged = np.zeros(k + 2, dtype=np.uint32)
for i, b in enumerate(bands):
    fw, ww, rr = b.segment_write(int(ffo[i]), i == 7, pkg.max_words(n, n // 8) // 3)
    merged[fw:fw + len(ww)] |= ww
    b.close()
print("bands == whole:", bool(np.array_equal(merged[:k], whole)), "result", int(rr["n_full_words"]) == int(res["n_full_words"]), flush=True)

t0 = time.time()
want, r = O.encode(img_d.cpu().numpy())
print("oracle %.1f s; identical: %s" % (time.time() - t0, bool(len(want) == k and np.array_equal(want, whole)
                                                              and r.tail_word == int(res["tail_word"]))), flush=True)
