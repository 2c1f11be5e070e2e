#!/usr/bin/env python3
"""Randomised GPU-vs-oracle comparison (run on a GPU box): python tools/fuzz_gpu.py [seconds] [seed]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

pkg = importlib.import_module("jpeg-encoder_amd")


def image(rng, h, w):
    kind = rng.integers(0, 7)
    if kind == 0:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == 1:
        return pkg.synth.natural_like(h, w, int(rng.integers(0, 1 << 30)))
    if kind == 2:
        return np.full((h, w, 3), rng.integers(0, 256, 3), dtype=np.uint8)
    if kind == 3:  # extremes in 8x8 patches
        a = rng.integers(0, 2, ((h + 7) // 8, (w + 7) // 8, 3)) * 255
        return np.kron(a, np.ones((8, 8, 1))).astype(np.uint8)[:h, :w]
    if kind == 4:  # pixel checkerboard of extremes: the longest codes
        y, x = np.mgrid[0:h, 0:w]
        return (((x + y) & 1) * 255)[..., None].repeat(3, 2).astype(np.uint8)
    if kind == 5:  # smooth with a noisy band (mixed tiers inside an image)
        img = pkg.synth.natural_like(h, w, int(rng.integers(0, 1 << 30)))
        r0 = int(rng.integers(0, h))
        img[r0:r0 + 16] = rng.integers(0, 256, img[r0:r0 + 16].shape, dtype=np.uint8)
        return img
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[rng.random((h, w)) < 0.7] = 255  # many all-ones bytes
    return img


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    t0, n, px = time.time(), 0, 0
    while time.time() - t0 < budget:
        kw = {}
        if rng.random() < 0.4:
            kw["qtabs"] = [rng.integers(1, int(rng.integers(2, 256)), 64) for _ in range(3)]
        kw["rgb_wrap"] = bool(rng.random() < 0.7)
        kw["cb_6d64"] = bool(rng.random() < 0.7)
        cfg = O.default_config(qtabs=kw.get("qtabs"), rgb_wrap=kw["rgb_wrap"], cb_6d64=kw["cb_6d64"])
        with pkg.JpegTop(**kw) as e:
            for _ in range(4):
                h, w = 8 * int(rng.integers(1, 40)), 8 * int(rng.integers(1, 70))
                nimg = int(rng.integers(1, 4))
                imgs = np.stack([image(rng, h, w) for _ in range(nimg)])
                words, res = e.encode_batch(imgs)
                for i in range(nimg):
                    want, r = O.encode(imgs[i], cfg)
                    ok = (len(words[i]) == len(want) and np.array_equal(words[i], want)
                          and int(res[i]["tail_bits"]) == r.tail_bits and int(res[i]["tail_word"]) == r.tail_word
                          and int(res[i]["total_bytes_stuffed"]) == r.total_bytes_stuffed)
                    if not ok:
                        np.save("gpurun_out/fuzz_fail.npy", imgs)
                        bad = np.nonzero(words[i][:min(len(want), len(words[i]))] != want[:min(len(want), len(words[i]))])[0]
                        print("len", len(words[i]), len(want), "first bad words", bad[:8], "of", len(bad))
                        print("MISMATCH", h, w, kw, "image", i)
                        sys.exit(1)
                    n += 1
                    px += h * w
    print("fuzz ok: %d images, %.1f Mpixel, %.0f s" % (n, px / 1e6, time.time() - t0))


if __name__ == "__main__":
    main()
