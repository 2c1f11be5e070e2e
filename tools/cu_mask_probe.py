#!/usr/bin/env python3
"""How the kernels of a step scale with the number of compute units (run on a GPU box):
    python tools/cu_mask_probe.py [frames]
Runs the 1080p batch on streams created with hipExtStreamCreateWithCUMask (a plain stream, every CU, halves and quarters
of the mask bits in several patterns) and prints the kernel times of each.  A kernel whose time doubles on half the
CUs is limited per CU (issue, latency) and gains nothing from a set of CUs of its own beside another kernel; one
whose time stays is limited by the memory system and would."""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("jpeg-encoder_amd")


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    w, h = 1920, 1080
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    frames = torch.empty((nf, h, w, 3), dtype=torch.uint8, device=dev)
    for s in range(0, nf, 16):
        e = min(nf, s + 16)
        frames[s:e] = pkg.synth.natural_like_torch(e - s, h, w, 1000 + s, dev)
    cap_words = (w * h * 3) // 4
    words = torch.zeros((nf, cap_words), dtype=torch.int32, device=dev)
    results = torch.zeros((nf, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    n_words = (n_cu + 31) // 32
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    masks = [("plain stream", None),
             ("all %d CUs" % n_cu, [0xFFFFFFFF] * n_words),
             ("every second pair of bits", [0x33333333] * n_words),
             ("every second four bits", [0x0F0F0F0F] * n_words),
             ("lower half of the bits", [0xFFFFFFFF] * (n_words // 2) + [0] * (n_words - n_words // 2)),
             ("low 16 of every 32 bits", [0x0000FFFF] * n_words),
             ("low 8 of every 32 bits", [0x000000FF] * n_words),
             ("low 24 of every 32 bits", [0x00FFFFFF] * n_words)]
    ref = None
    for name, m in masks:
        st = C.c_void_p()
        if m is None:
            rc = hip.hipStreamCreate(C.byref(st))
        else:
            arr = (C.c_uint32 * n_words)(*m)
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), n_words, arr)
        if rc != 0:
            print(name, "stream creation failed:", rc, flush=True)
            continue
        enc = pkg.JpegTop(device=0, stream=st.value)
        enc.reserve(nf, w, h)
        enc.set_profiling(True)
        for it in range(4):
            if it == 1:
                enc.synchronize()
                enc.kernel_times(reset=True)
            enc.encode_batch_device(frames.data_ptr(), nf, w, h, w * 3, h * w * 3, words.data_ptr(), cap_words,
                                    results.data_ptr())
        enc.synchronize()
        kt = enc.kernel_times(reset=True)
        torch.cuda.synchronize()
        chk = int(words.view(torch.int64).sum().item())
        if ref is None:
            ref = chk
        print("%-26s %s  output %s" % (name, {k: round(v[1] / 3, 3) for k, v in kt.items() if v[1]},
                                       "same" if chk == ref else "DIFFERENT"), flush=True)
        enc.close()
        hip.hipStreamDestroy(st)

    # Two contexts side by side, each on half of the mask bits, each coding all the frames: if the halves are
    # as fast together as alone, what a kernel loses on half the CUs is not a limit of the whole chip (clock, memory).
    import time
    halves = [[0xFFFFFFFF] * (n_words // 2) + [0] * (n_words - n_words // 2),
              [0] * (n_words // 2) + [0xFFFFFFFF] * (n_words - n_words // 2)]
    words2 = torch.zeros((nf, cap_words), dtype=torch.int32, device=dev)
    results2 = torch.zeros((nf, 32), dtype=torch.uint8, device=dev)
    encs, sts = [], []
    for m in halves:
        arr = (C.c_uint32 * n_words)(*m)
        st = C.c_void_p()
        assert hip.hipExtStreamCreateWithCUMask(C.byref(st), n_words, arr) == 0
        e = pkg.JpegTop(device=0, stream=st.value)
        e.reserve(nf, w, h)
        e.set_profiling(True)
        encs.append(e)
        sts.append(st)
    outs = [(words, results), (words2, results2)]
    for together in (False, True):
        for e in encs:
            e.kernel_times(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(3):
            for e, (wd, rs) in zip(encs, outs):
                e.encode_batch_device(frames.data_ptr(), nf, w, h, w * 3, h * w * 3, wd.data_ptr(), cap_words, rs.data_ptr())
                if not together:
                    e.synchronize()
        for e in encs:
            e.synchronize()
        dt = (time.perf_counter() - t0) / 3 * 1e3
        kts = [{k: round(v[1] / 3, 3) for k, v in e.kernel_times(reset=True).items() if v[1]} for e in encs]
        chk2 = int(words2.view(torch.int64).sum().item())
        print("two halves %-18s %.3f ms per 2 x %d frames  %s  output %s" % (
            "side by side" if together else "one after the other", dt, nf, kts, "same" if chk2 == ref else "DIFFERENT"), flush=True)


if __name__ == "__main__":
    main()
