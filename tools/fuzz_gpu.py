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
    last = t0
    while time.time() - t0 < budget:
        if time.time() - last > 20:
            last = time.time()
            print("  ... %d images, %.0f s" % (n, last - t0), flush=True)
        kw = {}
        if rng.random() < 0.4:
            kw["qtabs"] = [rng.integers(1, int(rng.integers(2, 256)), 64) for _ in range(3)]
        kw["rgb_wrap"] = bool(rng.random() < 0.7)
        kw["cb_6d64"] = bool(rng.random() < 0.7)
        cfg = O.default_config(qtabs=kw.get("qtabs"), rgb_wrap=kw["rgb_wrap"], cb_6d64=kw["cb_6d64"])
        with pkg.JpegTop(**kw) as e:
            # launch structure and LDS region size change the speed only: any combination must give the same words
            e.set_pipeline(int(rng.integers(0, 3)))
            e.set_region_mode(int(rng.integers(0, 4)))
            # chunk slots sized from the calls before (host-pointer calls repeat themselves when a chunk outgrows them)
            e.set_slot_mode(int(rng.integers(0, 2)))
            for _ in range(4):
                h, w = 8 * int(rng.integers(1, 40)), 8 * int(rng.integers(1, 70))
                big = rng.random() < 0.04
                if big:  # now and then a big one: thousands of chunks, long prefix sums
                    h, w = 8 * int(rng.integers(100, 140)), 8 * int(rng.integers(200, 260))
                mode = rng.integers(0, 5)
                if mode == 4:  # coefficients straight into the coders (any 11-bit values; the context's tables play no part)
                    hh, ww = (h, w) if not big else (8 * int(rng.integers(1, 40)), 8 * int(rng.integers(1, 70)))
                    nb = (hh // 8) * (ww // 8)
                    dens = float(rng.choice([0.02, 0.1, 0.4, 1.0]))
                    c = rng.integers(-1024, 1024, (nb, 3, 64)) * (rng.random((nb, 3, 64)) < dens)
                    c[rng.random((nb, 3, 64)) < 0.002] = -1024
                    c = c.astype(np.int16)
                    got, res = e.encode_coeffs(c, ww, hh)
                    want, r = O.encode_coeffs(c, O.default_config(cb_6d64=kw["cb_6d64"]))
                    if not (np.array_equal(got, want) and int(res["tail_bits"]) == r.tail_bits
                            and int(res["total_bytes_stuffed"]) == r.total_bytes_stuffed):
                        np.save("gpurun_out/fuzz_fail_coeffs.npy", c)
                        print("COEFFS MISMATCH", hh, ww, dens, kw)
                        sys.exit(1)
                    n += 1
                    px += hh * ww
                    continue
                if mode == 3 and h >= 16:  # one stream from bands (fresh contexts, same configuration)
                    img = image(rng, h, w)
                    cuts = sorted(set([0, h] + [8 * int(c) for c in rng.integers(1, h // 8, int(rng.integers(1, 4)))]))
                    bands = [pkg.JpegTop(**kw) for _ in cuts[:-1]]
                    info = [b.segment_encode(img, a, c - a) for b, a, c in zip(bands, cuts[:-1], cuts[1:])]
                    offs = np.concatenate([[0], np.cumsum([i[0] for i in info])])
                    ffs = [b.segment_count(int(offs[i]), info[i - 1][1] if i else 0) for i, b in enumerate(bands)]
                    ffo = np.concatenate([[0], np.cumsum(ffs)])
                    out = np.zeros(pkg.max_words(w, h), dtype=np.uint32)
                    for i, (b, a, c) in enumerate(zip(bands, cuts[:-1], cuts[1:])):
                        fw, ww, rr = b.segment_write(int(ffo[i]), i == len(bands) - 1, pkg.max_words(w, c - a) + 2)
                        out[fw:fw + len(ww)] |= ww
                        b.close()
                    want, r = O.encode(img, cfg)
                    if not (np.array_equal(out[:len(want)], want) and not out[len(want):].any()
                            and int(rr["n_full_words"]) == r.n_full_words and int(rr["tail_bits"]) == r.tail_bits):
                        np.save("gpurun_out/fuzz_fail.npy", img)
                        print("BAND MISMATCH", h, w, cuts, kw)
                        sys.exit(1)
                    n += 1
                    px += h * w
                    continue
                if mode == 2 and h >= 16:  # stripes of one image
                    img = image(rng, h, w)
                    divs = [d for d in range(1, h // 8 + 1) if (h // 8) % d == 0]
                    ns = int(rng.choice(divs))
                    words, res = e.encode_stripes(img, ns)
                    rows = h // ns
                    imgs = np.stack([img[rows * k:rows * (k + 1)] for k in range(ns)])
                    nimg = ns
                    h = rows
                else:
                    nimg = int(rng.integers(1, 4)) if not big else int(rng.integers(1, 9))  # (up to 4500 chunks)
                    imgs = None
                if imgs is None:
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
