#!/usr/bin/env python3
"""Benchmark of the jpeg_top hot path on MI355X: encoded Mpixel/s on 1080p RGB batches.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (for N > 1 the driver starts this file under torch.distributed.run).  A step is one
pass of the hot path -- RGB frames resident in HBM in, stuffed JPEG scan words resident in HBM out --
over this rank's batch of synthetic 1080p frames.  Weak scaling: every rank holds `--frames` frames
(default 1024; 8 ranks = the 8192-frame configuration), sharded by image, no data-path collective.
Rank 0 prints one JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--content", choices=["natural", "noise"], default="natural")
    ap.add_argument("--cpu-frames", type=int, default=32, help="frames of the batch timed on the CPU oracle")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--persistence", type=float, default=None,
                    help="octave amplitude ratio of the natural-like generator (default 0.7; larger = more detail)")
    ap.add_argument("--quant", type=int, default=1, help="every quantiser table entry (1: the RTL default)")
    ap.add_argument("--ws-mb", type=int, default=0, help="workspace limit in MiB (0: library default)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    return ap.parse_args()


def cpu_baseline(frames_np, gpu_words, gpu_res, n_threads, qtabs=None):
    """Times the CPU oracle (a scalar C++ port of the RTL arithmetic) on a bounded sample of the same
    frames, one frame per thread, and checks its output against the GPU's, byte for byte."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    O.lib()
    cfg = O.default_config(qtabs=qtabs)
    n = len(frames_np)
    # single thread on a quarter of the sample
    n1 = max(1, n // 4)
    t0 = time.perf_counter()
    for i in range(n1):
        O.encode(frames_np[i], cfg)
    t1 = time.perf_counter() - t0
    px = frames_np[0].shape[0] * frames_np[0].shape[1]
    single = n1 * px / t1 / 1e6
    out = [None] * n

    def work(k):
        for i in range(k, n, n_threads):
            out[i] = O.encode(frames_np[i], cfg)  # ctypes releases the GIL during the call

    threads = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    tm = time.perf_counter() - t0
    multi = n * px / tm / 1e6
    mismatches = 0
    for i in range(n):
        words, r = out[i]
        k = int(gpu_res[i]["n_full_words"]) + (1 if gpu_res[i]["tail_bits"] else 0)
        same = (k == len(words) and np.array_equal(gpu_words[i][:k], words)
                and int(gpu_res[i]["tail_bits"]) == r.tail_bits and int(gpu_res[i]["tail_word"]) == r.tail_word)
        mismatches += 0 if same else 1
    import shutil
    sim = next((t for t in ("verilator", "iverilog") if shutil.which(t)), None)
    return dict(value=round(multi, 2), unit="Mpixel/s", cores=n_threads, kind="port",
                rtl_sim="unavailable (no verilator/iverilog on this host; the RTL does not elaborate either)"
                if sim is None else "%s present, RTL not built (it does not elaborate)" % sim,
                host_cpus=os.cpu_count(),
                sample="%d frames of the timed batch, one frame per thread (C++ oracle, -O2)" % n,
                single_thread_mpixel_s=round(single, 2), seconds=round(t1 + tm, 2),
                frames_compared_with_gpu=n, mismatching_frames=mismatches)


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = "nccl"
    if os.environ.get("JPEGENC_BENCH_REHEARSAL") == "1":
        # rehearsal of the multi-rank path on a one-GPU box: every rank uses GPU 0, collectives over gloo
        local_rank, backend = 0, "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pkg = importlib.import_module("jpeg-encoder_amd")

    if args.persistence is not None:
        pkg.synth.PERSISTENCE = args.persistence
    w, h, nf = args.width, args.height, args.frames
    px_per_frame = w * h
    # this rank's shard: frames [rank*nf, (rank+1)*nf) of the global batch, seeds 1000 + global index
    frames = torch.empty((nf, h, w, 3), dtype=torch.uint8, device=dev)
    gen_chunk = 16
    for s in range(0, nf, gen_chunk):
        e = min(nf, s + gen_chunk)
        if args.content == "natural":
            frames[s:e] = pkg.synth.natural_like_torch(e - s, h, w, 1000 + rank * nf + s, dev)
        else:
            g = torch.Generator(device=dev)
            g.manual_seed(1000 + rank * nf + s)
            frames[s:e] = torch.randint(0, 256, (e - s, h, w, 3), generator=g, device=dev, dtype=torch.uint8)
    cap_words = (px_per_frame * (6 if args.content == "noise" else 3)) // 4  # 3 (6) bytes per pixel of room
    words = torch.zeros((nf, cap_words), dtype=torch.int32, device=dev)
    results = torch.zeros((nf, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    qtabs = None if args.quant == 1 else [[args.quant] * 64] * 3
    enc = pkg.JpegTop(qtabs=qtabs, device=local_rank, stream=torch.cuda.current_stream().cuda_stream)

    if args.ws_mb:
        enc.set_workspace_limit(args.ws_mb << 20)

    def step():
        enc.encode_batch_device(frames.data_ptr(), nf, w, h, w * 3, h * w * 3, words.data_ptr(), cap_words,
                                results.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    enc.kernel_times(reset=True)
    enc.set_profiling(not args.no_profile)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    enc.set_profiling(False)
    ktimes = enc.kernel_times(reset=True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    res = results.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(nf)
    ok = bool((res["status"] == 0).all())
    stuffed = int(res["total_bytes_stuffed"].sum())
    out_bytes_per_px = stuffed / float(nf * px_per_frame)
    total_px = float(world) * nf * px_per_frame * args.steps
    value = total_px / elapsed / 1e6

    line = {
        "metric": "encoded Mpixel/s (1080p RGB batch)", "value": round(value, 1), "unit": "Mpixel/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "%d x %dx%d RGB frames per GPU, %s content, %s quant tables; "
                               "8 GPUs = the 8192-frame 1080p configuration, sharded by image"
                               % (nf, w, h, ("natural-like" if args.persistence is None else
                                             "natural-like (persistence %.2f)" % args.persistence)
                                  if args.content == "natural" else "uniform-noise",
                                  "RTL default (all-ones)" if args.quant == 1 else "all-%d" % args.quant),
                   "frames_per_gpu": nf, "width": w, "height": h, "sharding": "independent images, no collective"},
        "all_images_ok": ok, "out_bytes_per_pixel": round(out_bytes_per_px, 4),
    }
    if rank == 0:
        # roofline of the dominant kernel, live HIP-event durations of this run
        alg_bytes_step = 3.0 * nf * px_per_frame + stuffed  # RGB read once + stuffed stream written once
        if not args.no_profile and ktimes["k_encode_chunks"][0] > 0:
            n_launch, ms_sum = ktimes["k_encode_chunks"]
            launches_per_step = n_launch / args.steps
            avg_ms = ms_sum / n_launch
            achieved = alg_bytes_step / launches_per_step / (avg_ms * 1e-3) / 1e9
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            valu_busy = None
            if os.path.exists(tfile):
                try:
                    pmc = json.load(open(tfile))
                    per_frame = pmc.get("k_encode_chunks_hbm_bytes_per_frame")
                    traffic = round(per_frame * nf / launches_per_step) if per_frame else None
                    valu_busy = pmc.get("valu", {}).get("k_encode_chunks_valu_busy")
                except Exception:
                    traffic = None
            line["roofline"] = {"bound": "hbm", "kernel": "k_encode_chunks", "achieved": round(achieved, 1),
                                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                                "traffic": traffic,
                                "algorithmic_bytes_per_launch": round(alg_bytes_step / launches_per_step),
                                "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step}
            if valu_busy is not None:  # counter pass kept under profiles/: the kernel is issue-bound, not HBM-bound
                line["roofline"]["valu_busy_from_profile"] = valu_busy
            all_ms = sum(v[1] for v in ktimes.values()) / args.steps
            line["kernel_ms_per_step"] = {k: round(v[1] / args.steps, 4) for k, v in ktimes.items()}
            line["pipeline_hbm_frac"] = round(alg_bytes_step / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        if not args.no_cpu:
            n_cpu = min(args.cpu_frames, nf)
            frames_np = frames[:n_cpu].cpu().numpy()
            k_max = int((res["n_full_words"][:n_cpu] + 1).max())
            gw = words[:n_cpu, :min(cap_words, k_max)].cpu().numpy().view(np.uint32)
            line["cpu_baseline"] = cpu_baseline(list(frames_np), gw, res[:n_cpu], args.cpu_threads, qtabs)
        print(json.dumps(line))
    enc.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
