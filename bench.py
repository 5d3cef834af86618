#!/usr/bin/env python3
"""bench.py — Mpixel/s of the MI355X HEIC decode hot path (BASELINE.json metric).

One "step" = one pass of the whole hot path over one batch of synthetic coded stills that are
already resident in HBM: CABAC parse -> intra/transform reconstruction -> deblock -> SAO+crop ->
fused YCbCr->RGB, all through the C ABI (include/heif_hipdec.h).  N=1 workload = BASELINE config 1
(3840x2160 4:2:0 8-bit stills, fused YCbCr->RGB24), `--batch` independent stills per step
(throughput form); the single-still latency form is reported beside it as `single_still`.
With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank decodes its own batch
(weak scaling, no data-path collective: independent stills do not exchange anything); the
`--workload grid` form shards the 48 tiles of an 8K grid over the ranks and gathers the decoded
tiles onto rank 0's canvas with RCCL (strong scaling).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="still4k", choices=["still4k", "still1080", "grid8k"])
    ap.add_argument("--batch", type=int, default=0, help="independent stills per rank and step (0 = workload default)")
    ap.add_argument("--qp", type=int, default=27)
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic contents cycled through the batch")
    ap.add_argument("--streams", type=int, default=1, help="independent sub-batches on separate HIP streams (the pipeline is instruction-issue bound: overlapping sub-batches gains nothing, measured 10.6 vs 11.0 Gpixel/s)")
    ap.add_argument("--enc", action="append", default=[], help="override a synthetic-encoder parameter, e.g. --enc wpp=0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the CPU baseline (0 = min(32, host cores))")
    return ap.parse_args()


WORKLOADS = {
    # name: (width, height, default batch, encoder config)
    "still4k": (3840, 2160, 1024, dict(wpp=1)),
    "still1080": (1920, 1080, 1024, dict(wpp=1)),
    "grid8k": (1024, 1024, 48, dict(wpp=1, vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)),
}


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from tools import streamgen

    w, h, def_batch, enc_cfg = WORKLOADS[a.workload]
    enc_cfg = dict(enc_cfg, qp=a.qp)
    for kv in a.enc:
        k, v = kv.split("=")
        enc_cfg[k] = int(v)
    grid = a.workload == "grid8k"
    if grid:
        total_items = 48
        items = [t for t in range(total_items) if t % world == rank]   # tile t -> GPU t mod G
        specs = [(w, h, 2 + t, 8, enc_cfg) for t in items]
    else:
        nb = a.batch or def_batch
        nd = max(1, min(a.distinct, nb))
        specs = [(w, h, 1 + rank * nd + i, 8, enc_cfg) for i in range(nd)]
    # ---- synthetic inputs (outside the timed region).  Generated in forked worker processes BEFORE this process touches
    #      the GPU or starts RCCL: forking a process that holds a HIP context / communicator threads is fragile ----
    distinct = streamgen.make_streams(specs)

    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decode path has no CPU fallback")
    torch.cuda.set_device(local_rank)

    import numpy as np
    import libheif_amd
    from libheif_amd.decoder import Batch
    from libheif_amd._capi import check

    lib = libheif_amd.load_library()
    check(lib.hipdec_init(local_rank))

    px_item = w * h
    gd = None
    if grid:
        from libheif_amd.grid import GridDecoder, GridLayout
        layout = GridLayout(6, 8, w, h, 8 * w, 6 * h)
        gd = GridDecoder(dict(zip(items, distinct)), layout, rank, world)   # tiles sharded t mod G, gather to rank 0
        batch = gd.batch
        streams = distinct
    else:
        streams = [distinct[i % len(distinct)] for i in range(nb)]
        batch = None
    n_items = len(streams)
    bs_bytes = sum(len(s) for s in streams)
    # the step's stills are split into `--streams` sub-batches, each one set of launches on its own HIP stream
    subs = []
    if grid:
        subs = [(batch, None)]
    else:
        k = max(1, min(a.streams, n_items))
        lib.hipdec_set_concurrent_batches(k)   # the sub-batches overlap on the GPU: they share the CABAC pool's wave slots
        for j in range(k):
            part = streams[j::k]
            b = Batch(part)          # parses headers on the host and uploads everything to HBM
            b.alloc_rgb(10)
            subs.append((b, lib.hipdec_stream_create() if k > 1 else None))
        batch = subs[0][0]
    for b, _ in subs:
        b.timing_slots(max(1, a.steps))
    single = Batch([streams[0]])
    single.alloc_rgb(10)

    def step(b):
        if gd is not None and b is batch:
            gd.decode()
            if rank == 0:
                gd.to_rgb((1, 13, 6, 1))
            return
        if b is single:
            b.run()
            b.to_rgb_all()
            return
        for sb, st in subs:
            sb.run(st)
            sb.to_rgb_all(st)

    def sync():
        for _, st in subs:
            if st:
                check(lib.hipdec_stream_synchronize(st))
        check(lib.hipdec_stream_synchronize(None))
        torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        step(batch)
    sync()
    if a.warmup > 0:
        for b, _ in subs:
            b.status()   # device-side decode errors are loud
    for b, _ in subs:
        b.timing_slots(max(1, a.steps))
    barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(batch)
    sync(); barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    for b, _ in subs:
        b.status()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / a.steps * 1e3
    total_px = px_item * (48 if grid else n_items * world)
    value = total_px / (elapsed / a.steps) / 1e6

    # ---- per-kernel device time over the timed region (HIP events on the launch stream) ----
    acc = dict(parse=0.0, recon=0.0, deblock=0.0, sao=0.0, total=0.0)
    for b, _ in subs:      # device time per kernel, summed over the sub-batches (they overlap in wall time)
        for k in range(a.steps):
            t = b.slot_timing_us(k)
            for key in acc:
                acc[key] += t[key]
    avg_us = {k: v / a.steps for k, v in acc.items()}

    # single-still latency form (one 4K still per pass)
    for _ in range(2):
        step(single)
    sync()
    ts = time.perf_counter()
    reps = 5
    for _ in range(reps):
        step(single)
    sync()
    single_ms = (time.perf_counter() - ts) / reps * 1e3
    single_t = single.timing_us()
    # the plugin life cycle on one still, host to host (new_decoder -> push_data -> decode -> D2H of the planes -> free):
    # what heif_decode_image() pays per item through libheif, PCIe included
    from libheif_amd.decoder import HipDecoder
    tp = time.perf_counter()
    for _ in range(3):
        dec = HipDecoder(); dec.push_data(streams[0]); dec.decode_next_image(); dec.free()
    plugin_ms = (time.perf_counter() - tp) / 3 * 1e3

    out = None
    if rank == 0:
        px_rank = px_item * n_items
        beta = bs_bytes / px_rank
        # coded-sample fraction of the workload (from the device unit maps of one still, outside the timed region):
        # the parser writes and the residual / reconstruction kernels read 2 B per coded sample
        m = single.maps(0)["flags"]
        coded = float((m & 1).mean() + 0.25 * ((m >> 1) & 1).mean() + 0.25 * ((m >> 2) & 1).mean())
        # algorithmic bytes per luma pixel, 8-bit samples:
        #   parse    beta (bitstream) in + 5/16 B unit maps + 2 B per coded sample (coefficient levels) out
        #   recon    (residual kernel + prediction wavefront) 2 x 2 B per coded sample in/out + 2 B in + 1.5 B out + maps
        #   deblock  3 B (1.5 read + 1.5 write) per pass, two passes (vertical + horizontal edges)
        #   sao      1.5 B in + 1.5 B out
        alg = dict(parse=beta + 5 / 16 + 2 * coded, recon=6 * coded + 1.5 + 5 / 16, deblock=2 * 3.0, sao=3.0)
        kernels = {}
        for k in ("parse", "recon", "deblock", "sao"):
            gbs = alg[k] * px_rank / (avg_us[k] * 1e-6) / 1e9 if avg_us[k] > 0 else 0.0
            kernels[k] = dict(avg_us=round(avg_us[k], 1), alg_bytes_per_px=round(alg[k], 4), achieved_gbs=round(gbs, 2),
                              frac=round(gbs / HBM_PEAK_GBS, 5))
        dom = max(("parse", "recon", "deblock", "sao"), key=lambda k: avg_us[k])
        # HBM traffic of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE separately) over this
        # same workload, recorded per luma pixel in profiles/pmc_traffic.json by tools/prof_hbm_traffic.sh
        traffic = None
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            kname = {"parse": "k_parse", "recon": "k_recon", "deblock": "k_deblock", "sao": "k_sao"}[dom]
            traffic = round(rec["bytes_per_px"][kname] * px_rank / len(subs), 0)
        except Exception:
            pass
        roofline = dict(bound="hbm", kernel={"parse": "k_parse", "recon": "k_residual+k_recon", "deblock": "k_deblock", "sao": "k_sao"}[dom],
                        achieved=round(kernels[dom]["achieved_gbs"] , 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=kernels[dom]["frac"], traffic=traffic,
                        launches_per_step=len(subs),
                        note="dominant kernel by device time (device times of the sub-batches are summed; they overlap in wall time); "
                             "CABAC parsing is bound by instruction issue (one dependency chain per substream, ~60 wave-instructions per pixel), not by HBM (DESIGN.md §4)")
        e2e_alg = (beta + 6.0) * px_rank   # drop-in end-to-end bytes: beta + 1.5 + 1.5 + 3
        out = {
            "metric": "Mpixels/s HEIC 4:2:0 8-bit decode", "value": round(value, 2), "unit": "Mpixel/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if grid else "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic (seeded noise+gradient stills coded by the test-only HEVC intra encoder, QP %d)" % a.qp,
            "config": {"workload": ("8K grid, 48 tiles of 1024x1024, tiles sharded over ranks" if grid else
                                    "%dx%d HEIC 4:2:0 8-bit stills, WPP, CTB 64, fused YCbCr->RGB24" % (w, h)),
                       "stills_per_step_per_gpu": n_items, "bitstream_bytes_per_px": round(beta, 4),
                       "substreams_per_still": batch.info(0)["num_substreams"], "hip_streams": len(subs),
                       "parallelism": "replicas x%d" % world},
            "roofline": roofline,
            "kernels": kernels,
            "end_to_end": {"alg_bytes_per_px": round(beta + 6.0, 3),
                           "achieved_gbs": round(e2e_alg / (elapsed / a.steps) / 1e9, 2),
                           "frac_of_hbm_peak": round(e2e_alg / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS, 5)},
            "single_still": {"ms": round(single_ms, 3), "mpixel_s": round(px_item / single_ms / 1e3, 2),
                             "kernel_us": {k: round(v, 1) for k, v in single_t.items()},
                             "plugin_lifecycle_host_to_host_ms": round(plugin_ms, 3)},
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(streams[0], px_item, a.cpu_seconds, a.cpu_procs)
    barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _cpu_worker(arg):
    stream, budget_s, max_n = arg
    from oracle import pyoracle as orc
    n = 0
    t0 = time.perf_counter()
    while True:
        r = orc.decode(stream)
        y, cb, cr = r["planes"]
        orc.color_420_to_rgb24(y, cb, cr, (1, 13, 6, 1))
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= max_n:
            break
    return n, time.perf_counter() - t0


def cpu_baseline(stream, px, budget_s, procs):
    """The CPU oracle (a scalar, spec-literal port: libde265 itself is not available here) on this host: `procs`
    independent processes, each decoding the same still of the bench workload (+ the reference's integer
    4:2:0->RGB24 op) for a bounded time; throughput = all decodes / the slowest process' time."""
    import multiprocessing as mp
    procs = procs or min(32, os.cpu_count() or 1)
    per_proc_s = max(1.0, budget_s / 2)      # ~2 x budget_s core-seconds per process pair keeps the run short
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_cpu_worker, [(stream, per_proc_s, 6)] * procs)
    n = sum(r[0] for r in res)
    dt = max(r[1] for r in res)
    return {"value": round(px * n / dt / 1e6, 2), "unit": "Mpixel/s", "cores": procs, "kind": "port",
            "sample": "%d processes x ~%d decode(s) of one still of the bench workload (CPU oracle decode + integer "
                      "4:2:0->RGB24), %.1f s wall, %.0f core-seconds" % (procs, n // procs, dt, sum(r[1] for r in res))}


if __name__ == "__main__":
    main()
