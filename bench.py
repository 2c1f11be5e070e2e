#!/usr/bin/env python3
"""Benchmark of the jpeg_top hot path on MI355X: encoded Mpixel/s on 1080p RGB batches.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (for N > 1 the driver starts this file under torch.distributed.run).  A step is one
pass of the hot path -- RGB frames resident in HBM in, stuffed JPEG scan words resident in HBM out --
over this rank's batch of synthetic 1080p frames.  Weak scaling by default: every rank holds `--frames` frames
(1024; 8 ranks = the 8192-frame configuration), sharded by image, no data-path collective;
`--scaling strong --total-frames 8192` holds the 8192 frames fixed and gives rank r the contiguous range
shard_range(8192, r, N).  `--mode stripes16k` / `--mode mono16k` time the single 16384 x 16384 image instead:
restart-interval stripes gathered on rank 0, or one stream merged from one band per rank.
Rank 0 prints one JSON line; the exit status is non-zero when any image failed or differs from the oracle.
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
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed steps (the library settles its LDS region size after five calls: see jpegenc_set_region_mode)")
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--total-frames", type=int, default=8192, help="frames of the whole job (strong scaling)")
    ap.add_argument("--mode", choices=["batch", "stripes16k", "mono16k"], default="batch")
    ap.add_argument("--side", type=int, default=16384, help="side of the square image of the 16k modes")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--content", choices=["natural", "noise"], default="natural")
    ap.add_argument("--cpu-frames", type=int, default=128, help="frames of the batch timed on the CPU oracle")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--persistence", type=float, default=None,
                    help="octave amplitude ratio of the natural-like generator (default 0.7; larger = more detail)")
    ap.add_argument("--quant", type=int, default=1, help="every quantiser table entry (1: the RTL default)")
    ap.add_argument("--ws-mb", type=int, default=0, help="workspace limit in MiB (0: library default)")
    ap.add_argument("--regions", choices=["auto", "small", "large", "short"], default="auto",
                    help="LDS staging regions of the encode kernel (jpegenc_set_region_mode)")
    ap.add_argument("--slots", choices=["adaptive", "worst"], default="adaptive",
                    help="scratch per chunk: sized from the chunk lengths of earlier calls, or the worst case (jpegenc_set_slot_mode)")
    ap.add_argument("--split", action="store_true", help="separate kernels also for small calls (jpegenc_set_pipeline)")
    ap.add_argument("--sync-each-step", action="store_true",
                    help="wait for every step before the next is queued: the latency of one call, not the rate of a queue of them")
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
    elif args.mode != "batch":
        # the 16k modes run the exchange code with one rank too: a group of one
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, **kw)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pkg = importlib.import_module("jpeg-encoder_amd")

    if args.persistence is not None:
        pkg.synth.PERSISTENCE = args.persistence
    if args.mode != "batch":
        rc = bench_16k(args, pkg, torch, dist, dev, world, rank, backend)
        dist.barrier()
        dist.destroy_process_group()
        return rc
    w, h = args.width, args.height
    if args.scaling == "strong":
        first, last = pkg.distributed.shard_range(args.total_frames, rank, world)
        nf = last - first
    else:
        nf, first = args.frames, rank * args.frames
    total_frames = args.total_frames if args.scaling == "strong" else world * nf
    px_per_frame = w * h
    # this rank's shard: frames [first, first + nf) of the global batch, seeds 1000 + global index
    frames = torch.empty((nf, h, w, 3), dtype=torch.uint8, device=dev)
    gen_chunk = 16
    for s in range(0, nf, gen_chunk):
        e = min(nf, s + gen_chunk)
        if args.content == "natural":
            frames[s:e] = pkg.synth.natural_like_torch(e - s, h, w, 1000 + first + s, dev)
        else:
            g = torch.Generator(device=dev)
            g.manual_seed(1000 + first + s)
            frames[s:e] = torch.randint(0, 256, (e - s, h, w, 3), generator=g, device=dev, dtype=torch.uint8)
    cap_words = (px_per_frame * (6 if args.content == "noise" else 3)) // 4  # 3 (6) bytes per pixel of room
    words = torch.zeros((nf, cap_words), dtype=torch.int32, device=dev)
    results = torch.zeros((nf, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    qtabs = None if args.quant == 1 else [[args.quant] * 64] * 3
    enc = pkg.JpegTop(qtabs=qtabs, device=local_rank, stream=torch.cuda.current_stream().cuda_stream)

    if args.ws_mb:
        enc.set_workspace_limit(args.ws_mb << 20)
    enc.set_region_mode({"auto": enc.REGIONS_AUTO, "small": enc.REGIONS_SMALL, "large": enc.REGIONS_LARGE,
                         "short": enc.REGIONS_SHORT}[args.regions])
    if args.split:
        enc.set_pipeline(enc.PIPELINE_SPLIT)
    if args.slots == "adaptive":
        # the first call sizes the chunk slots from a first few frames (and waits for them, once); the warm-up
        # below brings the workspace to its final size, so the timed steps only enqueue work
        enc.set_slot_mode(enc.SLOTS_ADAPTIVE)
    else:
        enc.reserve(nf, w, h)  # the steps below only enqueue work

    def step():
        enc.encode_batch_device(frames.data_ptr(), nf, w, h, w * 3, h * w * 3, words.data_ptr(), cap_words,
                                results.data_ptr())
        if args.sync_each_step:
            enc.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1 if args.slots == "adaptive" else 0)):
        step()
        if args.slots == "adaptive":
            enc.synchronize()  # (statistics of this call in before the next one sizes its slots)
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
    unstuffed = int(res["total_bits_unstuffed"].sum()) // 8
    out_bytes_per_px = stuffed / float(nf * px_per_frame)
    total_px = float(total_frames) * px_per_frame * args.steps
    value = total_px / elapsed / 1e6
    if world > 1:  # a failed image on any rank fails the run
        flag = torch.tensor([0 if ok else 1], dtype=torch.int64, device=dev)
        dist.all_reduce(flag)
        ok = int(flag.item()) == 0

    content = (("natural-like" if args.persistence is None else "natural-like (persistence %.2f)" % args.persistence)
               if args.content == "natural" else "uniform-noise")
    tables = "RTL default (all-ones)" if args.quant == 1 else "all-%d" % args.quant
    if args.scaling == "strong":
        workload = ("%d x %dx%d RGB frames in all, %s content, %s quant tables, sharded by image over %d GPU(s): "
                    "%d frames on this GPU" % (total_frames, w, h, content, tables, world, nf))
    else:
        workload = ("%d x %dx%d RGB frames per GPU, %s content, %s quant tables; "
                    "8 GPUs = the 8192-frame 1080p configuration, sharded by image" % (nf, w, h, content, tables))
    line = {
        "metric": "encoded Mpixel/s (1080p RGB batch)", "value": round(value, 1), "unit": "Mpixel/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": workload, "frames_per_gpu": nf, "total_frames": total_frames, "width": w, "height": h,
                   "sharding": "independent images, no collective"},
        "all_images_ok": ok, "out_bytes_per_pixel": round(out_bytes_per_px, 4),
        "workspace": {"slots": args.slots, "bytes": enc.workspace_bytes(),
                      "bytes_per_frame": round(enc.workspace_bytes() / float(nf))},
    }
    over_small, over_large, counted, over_short, auto_at = enc.region_stats()
    if counted:  # chunks of the last counted launch that hold an MCU too long for a short / small / large LDS region
        line["chunks_over_region"] = {"short": round(over_short / counted, 4), "small": round(over_small / counted, 4),
                                      "large": round(over_large / counted, 4)}
        line["regions"] = args.regions if args.regions != "auto" else \
            "auto: " + {enc.REGIONS_SMALL: "small", enc.REGIONS_LARGE: "large", enc.REGIONS_SHORT: "short"}.get(auto_at, "?")
    failed = not ok
    if rank == 0:
        # Rooflines from the live HIP-event durations of this run (events recorded by the library on its stream).
        rgb_bytes = 3.0 * nf * px_per_frame
        pipe_bytes = rgb_bytes + stuffed   # the path's algorithmic traffic: RGB read once + stuffed stream written once
        enc_bytes = rgb_bytes + unstuffed  # what the dominant kernel itself moves: RGB in, unstuffed chunk streams out
        if not args.no_profile and ktimes["k_encode_chunks"][0] > 0:
            n_launch, ms_sum = ktimes["k_encode_chunks"]
            launches_per_step = n_launch / args.steps
            avg_ms = ms_sum / n_launch
            per_launch = enc_bytes / launches_per_step
            achieved = per_launch / (avg_ms * 1e-3) / 1e9
            all_ms = sum(v[1] for v in ktimes.values()) / args.steps
            pipe_achieved = pipe_bytes / (all_ms * 1e-3) / 1e9
            line["roofline"] = {
                "bound": "hbm", "kernel": "k_encode_chunks", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                "algorithmic_bytes_per_launch": round(per_launch),
                "algorithmic_bytes": "RGB bytes in + unstuffed chunk-stream bytes out of the frames of one launch",
                "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step,
                "pipeline": {"kernel": "all four kernels of a step", "achieved": round(pipe_achieved, 1),
                             "frac": round(pipe_achieved / HBM_PEAK_GBPS, 4),
                             "algorithmic_bytes_per_step": round(pipe_bytes),
                             "algorithmic_bytes": "RGB bytes in + stuffed stream bytes out",
                             "kernel_ms_per_step": round(all_ms, 4)},
            }
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tfile):  # counters are not collected in this run: figures of the stored counter passes
                try:
                    pmc = json.load(open(tfile))
                    per_frame = pmc.get("k_encode_chunks_hbm_bytes_per_frame")
                    valu = pmc.get("valu", {})
                    ipw = valu.get("valu_instructions_per_wave")
                    built = enc.lib.jpegenc_version().decode().split()[-1]
                    line["roofline"]["from_stored_profile"] = {
                        "source": "profiles/pmc_traffic.json (%s)" % pmc.get("build", "rocprofv3 --pmc passes"),
                        # which kernel sources the counters were collected on, and whether this run's library is that build
                        "profile_commit": pmc.get("commit"), "profile_kernels_sha": pmc.get("kernels_sha"),
                        "library_kernels_sha": built, "stale": pmc.get("kernels_sha") != built,
                        "traffic": round(per_frame * nf / launches_per_step) if per_frame else None,
                        "valu_insts_per_wave": ipw,
                        "lane_ops_per_pixel": round(ipw * 64 / 4096.0, 1) if ipw else None,
                        "valu_busy": valu.get("k_encode_chunks_valu_busy"),
                        "note": "the kernel is VALU-issue-bound; HBM is not its limit"}
                except Exception:
                    pass
            line["kernel_ms_per_step"] = {k: round(v[1] / args.steps, 4) for k, v in ktimes.items()}
        if not args.no_cpu:
            n_cpu = min(args.cpu_frames, nf)
            frames_np = frames[:n_cpu].cpu().numpy()
            k_max = int((res["n_full_words"][:n_cpu] + 1).max())
            gw = words[:n_cpu, :min(cap_words, k_max)].cpu().numpy().view(np.uint32)
            line["cpu_baseline"] = cpu_baseline(list(frames_np), gw, res[:n_cpu], args.cpu_threads, qtabs)
            failed = failed or line["cpu_baseline"]["mismatching_frames"] != 0
        if failed:
            line["error"] = "an image did not fit its output buffer or differs from the oracle: value is not valid"
        print(json.dumps(line))
    enc.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 1 if failed else 0


def bench_16k(args, pkg, torch, dist, dev, world, rank, backend):
    """The single huge image: `--mode stripes16k` codes it as 8 restart-interval stripes (stripe s on rank
    s % world) and gathers table + payloads on rank 0; `--mode mono16k` codes one band per rank and merges the
    bands into the one stream jpeg_top gives for the whole image.  A step is the whole job: coding plus the
    exchange.  Stripe s is natural-like content of seed 7 + s, generated where it is needed."""
    import numpy as np
    n, stripes = args.side, 8
    rows = n // stripes
    if n % (8 * stripes) or stripes % world:
        raise SystemExit("--side must be a multiple of 64 and the rank count must divide 8")
    xdev = dev if backend == "nccl" else torch.device("cpu")  # gloo rehearsal: the exchange goes through host memory
    enc = pkg.JpegTop(device=dev.index, stream=torch.cuda.current_stream().cuda_stream)

    def stripe_pixels(s):
        return pkg.synth.natural_like_torch(1, rows, n, 7 + s, dev)[0]

    if args.mode == "stripes16k":
        mine = pkg.distributed.stripes_of_rank(stripes, rank, world)
        local_img = torch.cat([stripe_pixels(s) for s in mine])  # this rank's stripes, one under the other
        if backend == "nccl":
            # one call for all of the rank's stripes; their records reach the host once, gathered
            resident = pkg.distributed.HipStripes(enc, local_img, len(mine))

            def step():
                return pkg.distributed.encode_striped_image(resident, stripes, device=xdev)
        else:
            coder = pkg.distributed.hip_stripe_coder(enc, local_img, len(mine))

            def step():
                local = {}
                for k, s in enumerate(mine):
                    wds, res = coder(k)
                    local[s] = (wds.to(xdev), res)
                return pkg.distributed.gather_stripes(local, stripes, device=xdev)
    else:
        per = stripes // world
        parts = [stripe_pixels(s) for s in range(rank * per, (rank + 1) * per)]
        if rank:
            parts.insert(0, stripe_pixels(rank * per - 1)[-8:])  # the 8 pixel rows above the band: DC predictors
        local_img = torch.cat(parts)
        lo = 8 if rank else 0

        class Band(pkg.distributed.HipBand):
            def write(self, ff_offset, is_last):
                fw, wds, res = super().write(ff_offset, is_last)
                return fw, wds.to(xdev), res

        def step():
            if backend == "nccl":  # figures and words stay on the GPU; the host waits once, for the gathered spans
                return pkg.distributed.encode_monolithic(
                    pkg.distributed.HipBandResident(enc, local_img, lo, lo + per * rows), device=xdev)
            return pkg.distributed.encode_monolithic(Band(enc, local_img, lo, lo + per * rows), device=xdev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rc = 0
    if rank == 0:
        if args.mode == "stripes16k":
            table, payload = out
            # the container: offsets are the running sum of the stripes' word counts, payload holds them all
            sizes = table[:, 6]
            good = bool((table[:, 0] == np.concatenate([[0], np.cumsum(sizes)[:-1]])).all()
                        and int(sizes.sum()) == payload.numel()
                        and (table[:, 1] + (table[:, 2] != 0) == sizes).all())
            out_bytes = int(table[:, 5].sum())
            extra = {"stripes": stripes, "container_consistent": good, "payload_words": int(payload.numel())}
        else:
            wds, res = out
            good = int(res["status"]) == 0 and int(res["n_full_words"]) + (1 if res["tail_bits"] else 0) <= wds.numel()
            out_bytes = int(res["total_bytes_stuffed"])
            extra = {"bands": world, "stream_words": int(res["n_full_words"])}
        if not args.no_cpu:  # one stripe against the oracle (2 s of CPU): stripe 0 is also band 0's first rows
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            t0 = time.perf_counter()
            want, r = O.encode(local_img[:rows].cpu().numpy()) if args.mode == "stripes16k" else (None, None)
            if want is not None:
                got = payload[:len(want)].cpu().numpy().view(np.uint32)
                good = good and np.array_equal(got, want) and int(table[0, 3]) == r.tail_word
                extra["cpu_baseline"] = {"value": round(rows * n / (time.perf_counter() - t0) / 1e6, 2), "unit": "Mpixel/s",
                                         "cores": 1, "kind": "port", "sample": "stripe 0 (%d x %d), compared word for word" % (n, rows)}
        if not args.no_cpu and args.mode == "mono16k" and n <= 8192:
            # the merged stream against the same image coded in one call on this GPU (bit-exact against the oracle
            # in tests/): every word and the end-of-file fields
            whole = torch.cat([stripe_pixels(s) for s in range(stripes)])
            cap = n * n * 3 // 4
            d_words = torch.zeros(cap, dtype=torch.int32, device=dev)
            d_res = torch.zeros(32, dtype=torch.uint8, device=dev)
            enc.encode_batch_device(whole.data_ptr(), 1, n, n, n * 3, n * n * 3, d_words.data_ptr(), cap, d_res.data_ptr())
            enc.synchronize()
            one = d_res.cpu().numpy().view(pkg.RESULT_DTYPE)[0]
            k = int(one["n_full_words"]) + (1 if one["tail_bits"] else 0)
            same = (int(one["status"]) == 0 and k <= wds.numel() and torch.equal(wds[:k].to(dev), d_words[:k])
                    and not bool(wds[k:].any())
                    and all(int(res[f]) == int(one[f]) for f in ("n_full_words", "tail_bits", "tail_word", "total_bytes_stuffed")))
            good = good and same
            extra["same_as_one_call"] = bool(same)
        rc = 0 if good else 1
        line = {"metric": "encoded Mpixel/s (one %dx%d image, %s)" % (n, n, args.mode), "value": round(n * n * args.steps / elapsed / 1e6, 1),
                "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "int32", "data": "synthetic",
                "config": {"workload": "one %dx%d natural-like RGB image, %s, exchange over %s" % (
                    n, n, "8 restart-interval stripes gathered on rank 0" if args.mode == "stripes16k"
                    else "one band per rank merged into jpeg_top's single stream", backend)},
                "out_bytes_per_pixel": round(out_bytes / float(n * n), 4), "result_ok": bool(good)}
        line.update(extra)
        print(json.dumps(line))
    enc.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
