#!/usr/bin/env python3
"""bench.py — Mpixel/s of the MI355X HEIC decode hot path (BASELINE.json metric).

One "step" = one pass of the whole hot path over one batch of synthetic coded stills: CABAC parse -> dequantisation /
inverse transforms -> intra reconstruction -> deblock -> SAO + crop -> fused YCbCr->RGB, all through the C ABI
(include/heif_hipdec.h).  N=1 workload = BASELINE config 1 (3840x2160 4:2:0 8-bit stills, fused YCbCr->RGB24), `--batch`
independent stills per step (throughput form).  Two timed regions are measured, W warm-up + K timed steps each:

  value / ms_per_step   SURVEY.md 8(d)'s definition, "from compressed bytes in host memory to planes + RGB complete in HBM" (round 4: the
                        headline, as the round-3 review asked): every step pays hipdec_batch_create_recycling (host header parsing on
                        worker threads, pinned staging, the H2D upload) + decode + colour stage, double-buffered against the previous
                        step's kernels;
  value_resident        inputs resident in HBM when the timed region starts: hipdec_batch_run_rgb per step (the per-kernel HIP-event
                        times and the roofline come from this region; with --only-main it is the only one and `value` reports it).

Beside them: per-kernel device times (HIP events on the launch stream) with 8(d)'s algorithmic bytes, the issue roofline of the
instruction-bound kernels, the batch-size curve (64 .. 2048 stills per batch), BASELINE.json's configs as written (`baseline_configs`:
one 4K still; the 8K grid; Main10 4K with the PQ -> linear-RGB stage inside the timed region; 1024 x 1080p), further synthetic inputs
(`extra_workloads`), the drop-in form through the real libheif, and the CPU oracle on the host cores.

With --gpus N (torch.distributed.run, one rank per GPU) every rank decodes its own batches (weak scaling, no data-path collective:
independent stills exchange nothing), and the line carries `grid_sharded`: ONE 8K grid photo (BASELINE config 3) with its 48 tiles
sharded t mod N over the N GPUs, once through hipdec_grid_* (rank 0 drives all devices, strided peer-copy paste) and once through
hipdec_grid_*_rccl (every rank decodes its tiles, grouped ncclSend / ncclRecv gather inside libheifhip.so), both checked against the
one-GPU canvas, with the strong-scaling efficiency stated.  `--workload grid8k` makes that single photo the main measurement (strong
scaling, as in rounds 1 - 2); `--workload grid8k_multi` gives every rank its own photos (weak).

After the timed regions the run CHECKS results: planes and RGB24 of 32 stills spread over the batch are read back and compared with the
CPU oracle's decode of their streams (the cpu_baseline leg decodes them anyway); a mismatch fails the run.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

# HIP reads this when the runtime initialises - at the process's first HIP call, which in this script is torch's, before libheifhip.so (whose load-time
# hook sets the same default for C hosts) is loaded: the chains of sequence tracks keep several launch sets in flight on streams of their own, and HIP's
# default of 4 hardware queues serialises streams that share one (libheif_amd/csrc/runtime.hip, profiles/r06_sequence_pipeline.txt)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

KERNEL_KEYS = ("parse", "residual", "recon", "deblock", "sao", "colour")
KERNEL_NAMES = {"parse": "k_parse", "residual": "k_residual", "recon": "k_recon", "deblock": "k_deblock", "sao": "k_sao", "colour": "k_ycbcr_to_rgb",
                "sao_rgb": "k_sao_rgb"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="still4k", choices=["still4k", "still1080", "main10_4k", "grid8k", "grid8k_multi"])
    ap.add_argument("--batch", type=int, default=0, help="independent stills per rank and step (0 = workload default)")
    ap.add_argument("--grid-single", action="store_true", help="(kept for old command lines: `--workload grid8k` IS the single-photo form again)")
    ap.add_argument("--no-grid-sharded", action="store_true", help="skip the grid_sharded section (one 8K grid photo over the N GPUs: peer copies and RCCL)")
    ap.add_argument("--verify-stills", type=int, default=32, help="stills of the batch checked against the CPU oracle after the timed regions")
    ap.add_argument("--no-dropin", action="store_true", help="skip the T threads x heif_decode_image() measurement through the real libheif + plugin")
    ap.add_argument("--qp", type=int, default=27)
    ap.add_argument("--distinct", type=int, default=256, help="distinct synthetic contents cycled through the batch")
    ap.add_argument("--enc", action="append", default=[], help="override a synthetic-encoder parameter, e.g. --enc wpp=0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the batch curve, BASELINE's other configs and the extra workloads")
    ap.add_argument("--only-main", action="store_true", help="main resident measurement only (profiling runs)")
    ap.add_argument("--only-sequences", action="store_true", help="the sequence_tracks section only (development)")
    ap.add_argument("--parts", type=int, default=1, help="batches a step is split into; with 2, batch k+1's CABAC parse is queued beside batch k's "
                    "pixel stages on a second stream (hipdec_set_stage_overlap).  Measured SLOWER (12.1 against 13.7 Gpixel/s): the CABAC work pool "
                    "holds every wave slot of the chip, the other kernels only start when it exits, and smaller batches parse less efficiently")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the CPU baseline (0 = min(32, host cores))")
    return ap.parse_args()


GRID_VUI = dict(wpp=1, vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
WORKLOADS = {
    # name: (width, height, default batch, bit depth, encoder config, output heif_chroma)
    "still4k": (3840, 2160, 2048, 8, dict(wpp=1), 10),
    "still1080": (1920, 1080, 1024, 8, dict(wpp=1), 10),
    "main10_4k": (3840, 2160, 256, 10, dict(wpp=1, vui_matrix=9, vui_primaries=9, vui_transfer=16), 14),
    "grid8k": (1024, 1024, 48, 8, GRID_VUI, 10),
    "grid8k_multi": (1024, 1024, 48, 8, GRID_VUI, 10),
    # 4:4:4 to RGB24 through Op_YCbCr_to_RGB + Op_RGB_to_RGB24_32 (no fused form); 4:2:2 Main10 (what cameras write) as planes: output chroma None
    "still1080_444": (1920, 1080, 512, 8, dict(wpp=1, chroma_format_idc=3), 10),
    "still1080_422_10": (1920, 1080, 512, 10, dict(wpp=1, chroma_format_idc=2), None),
}


def gen_streams(specs, rank=0, world=1):
    """Synthetic inputs (outside every timed region), generated in forked worker processes BEFORE this process touches the GPU
    or starts RCCL.  With several ranks each one codes a share of the set and picks the rest up from the on-disk cache."""
    from tools import streamgen
    if world > 1:
        streamgen.make_streams([sp for i, sp in enumerate(specs) if i % world == rank], workers=max(1, (os.cpu_count() or 2) // world))
        deadline = time.time() + 900
        while True:
            missing = [sp for sp in specs if not os.path.exists(streamgen.stream_path(*sp))]
            if not missing:
                break
            if time.time() > deadline:
                raise SystemExit("bench.py: synthetic streams of another rank did not appear")
            time.sleep(0.2)
    return streamgen.make_streams(specs)


class Workload:
    """n coded stills (cycled from `distinct` contents) as `parts` hipdec batches + one RGB output buffer per still."""

    def __init__(self, lib, name, distinct, n, out_chroma, w, h, bit_depth, parts=1):
        self.lib, self.name, self.out_chroma = lib, name, out_chroma
        self.w, self.h, self.bit_depth = w, h, bit_depth
        self.streams = [distinct[i % len(distinct)] for i in range(n)]
        self.n = n
        self.bs_bytes = sum(len(s) for s in self.streams)
        self.px = w * h * n
        self.parts = max(1, min(parts, n))
        self.part_streams = [self.streams[j::self.parts] for j in range(self.parts)]
        self.batches = []

    @property
    def batch(self):
        return self.batches[0] if self.batches else None

    def make_resident(self):
        from libheif_amd.decoder import Batch
        self.batches = []
        for st in self.part_streams:
            b = Batch(st)                       # host header parsing + upload: outside the resident form's timed region
            if self.out_chroma is not None:
                b.alloc_rgb(self.out_chroma)
            self.batches.append(b)
        return self.batches

    def step_resident(self):
        for b in self.batches:
            if self.out_chroma is None:
                b.run()        # planes only
            else:
                b.run_rgb()    # decode + colour stage (fused into the SAO store path for 8-bit 4:2:0 -> RGB24)

    def status(self):
        for b in self.batches:
            b.status()

    def timing_slots(self, n):
        for b in self.batches:
            b.timing_slots(n)

    def free(self):
        for b in self.batches:
            b.free()
        self.batches = []


def kernel_times(batches, steps):
    """device time per kernel and step, summed over the step's batches (with stage overlap they run beside each other: the sum of the
    kernel times then exceeds the step time)"""
    acc = {k: 0.0 for k in KERNEL_KEYS + ("decode_total",)}
    for b in batches:
        for s in range(steps):
            t = b.slot_kernel_timing_us(s)
            for k in acc:
                acc[k] += t[k]
    return {k: v / steps for k, v in acc.items()}


def coded_fraction(batch, chroma_weight=0.25):
    """coded samples per luma pixel (luma + 2 x chroma at chroma_weight samples per luma pixel: 0.25 / 0.5 / 1 for 4:2:0 / 4:2:2 / 4:4:4) from the
    device unit maps of item 0 (4:2:2: a unit's flags describe the upper one of its two chroma blocks)"""
    m = batch.maps(0)["flags"]
    return float((m & 1).mean() + chroma_weight * ((m >> 1) & 1).mean() + chroma_weight * ((m >> 2) & 1).mean())


def alg_bytes(beta, coded, s, s_out, spp=1.5):
    """algorithmic bytes per luma pixel: s = bytes per sample, s_out = bytes per output sample
         parse     beta (bitstream) in + 5/16 B unit maps + 2 B per coded sample (coefficient levels) out
         residual  2 B per coded sample in + 2 B out (in place)
         recon     1.5 s written + 2 B per coded sample read                              (8d "recon")
         deblock   3 s + 1/16 (bS / QP metadata), both edge directions together          (8d "deblock")
         sao       1.5 s in + 1.5 s out                                                   (8d "SAO")
         colour    1.5 s in + 3 s_out out                                                 (8d "fused colour stage")"""
    # spp = samples per luma pixel over the three planes: 1.5 (4:2:0; the figures above), 2 (4:2:2), 3 (4:4:4)
    return dict(parse=beta + 5 / 16 + 2 * coded, residual=4 * coded, recon=spp * s + 2 * coded, deblock=2 * spp * s + 1 / 16,
                sao=2 * spp * s, colour=spp * s + 3 * s_out)


def kernel_table(avg_us, alg, px, colour_stage=True):
    out = {}
    keys = list(KERNEL_KEYS)
    if not colour_stage:
        keys = [k for k in keys if k != "colour"]          # planes only: no colour stage ran, the SAO kernel is the plain one
    elif avg_us["colour"] == 0.0 and avg_us["sao"] > 0.0:
        # the colour stage ran inside the SAO kernel (k_sao_rgb): one pass reads the deblocked picture (1.5 s) and writes the planes (1.5 s) and
        # the interleaved RGB (3 s_out); the separate colour pass's re-read of the planes is gone
        keys = [k for k in keys if k not in ("sao", "colour")] + ["sao_rgb"]
        avg_us = dict(avg_us, sao_rgb=avg_us["sao"])
        alg = dict(alg, sao_rgb=alg["sao"] + alg["colour"] - (alg["sao"] / 2))
    for k in keys:
        gbs = alg[k] * px / (avg_us[k] * 1e-6) / 1e9 if avg_us[k] > 0 else 0.0
        out[k] = dict(kernel=KERNEL_NAMES[k], avg_us=round(avg_us[k], 1), alg_bytes_per_px=round(alg[k], 4), achieved_gbs=round(gbs, 2),
                      frac=round(gbs / HBM_PEAK_GBS, 5))
    return out


# What one gfx950 CU issues per cycle, measured at 8 waves per SIMD (tools/ubench/issue_model.hip, profiles/r06_issue_model_*.txt): the CU-shared scalar
# pipe 0.92 SALU instructions; the four SIMDs together 1.76 wave64 VALU instructions that touch only VGPRs / inline constants, but 0.90 of those that
# read or write the scalar register file (an SGPR operand, VCC, v_readlane / v_writelane, every v_cmp and v_cndmask).
ISSUE_PEAK_SALU, ISSUE_PEAK_VALU_PURE, ISSUE_PEAK_VALU_SFILE = 0.92, 1.76, 0.90


def issue_roofline(kernels, px, cu_count, clock_ghz):
    """The CABAC parser and the intra reconstruction are bound by instruction issue and per-wave latency, not by HBM.  Their
    yardsticks are the CU's issue rates as MEASURED in round 6 (constants above): the scalar pipe, and the vector pipes - whose rate depends on whether
    an instruction touches the scalar register file, which the PMC counters do not split, so the vector fraction is a RANGE: [all instructions pure,
    all instructions scalar-file].  (Rounds 2 - 5 priced the vector pipes at 2 per cycle and CU and read k_parse as scalar-bound at 45 % vector
    utilisation; measured, its 24 VALU per pixel ran at 0.906 per cycle and CU - the scalar-file rate.)  Instructions per luma pixel come from the
    committed PMC passes (profiles/pmc_issue.json, tools/prof_parse_pmc.sh); they are a property of the code and the content, not of the batch size."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_issue.json")))
    except Exception:
        return None
    eff = rec.get("effective_clock_ghz") or {}
    nominal = clock_ghz
    if eff.get("k_parse"):
        clock_ghz = float(eff["k_parse"])
    cyc = cu_count * clock_ghz            # G CU-cycles / s
    out = {"peaks_per_cycle_and_cu": {"salu": ISSUE_PEAK_SALU, "valu_vgpr_only": ISSUE_PEAK_VALU_PURE, "valu_touching_scalar_file": ISSUE_PEAK_VALU_SFILE},
           "peaks_source": "tools/ubench/issue_model.hip at 8 waves per SIMD on this chip (profiles/r06_issue_model_a.txt, _b.txt)",
           "peak_ginst_s": round(cyc * ISSUE_PEAK_SALU, 1), "clock_ghz": round(clock_ghz, 3),
           "clock_source": ("GRBM_GUI_ACTIVE / kernel time of k_parse in the PMC pass (profiles/pmc_issue.json)" if eff.get("k_parse")
                            else "nominal maximum clock %.2f GHz: the effective clock under load is lower, so the fractions are lower bounds" % nominal),
           "source": rec.get("source"), "commit": rec.get("commit"), "kernels": {}}
    for key, name in (("parse", "k_parse"), ("recon", "k_recon"), ("residual", "k_residual"), ("deblock", "k_deblock"), ("sao_rgb", "k_sao"), ("sao", "k_sao")):
        if key in kernels and key not in out["kernels"] and name in rec.get("insts_per_px", {}) and kernels[key]["avg_us"] > 0:
            ipp = rec["insts_per_px"][name]
            per_s = px / (kernels[key]["avg_us"] * 1e-6) / 1e9      # G pixels / s
            salu, valu = ipp["salu"] * per_s, ipp["valu"] * per_s
            out["kernels"][key] = {"kernel": kernels[key]["kernel"], "salu_per_px": ipp["salu"], "valu_per_px": ipp["valu"], "branch_per_px": ipp.get("branch"),
                                   "lds_per_px": ipp.get("lds"), "achieved_ginst_s": round(salu, 1), "frac_of_scalar_issue_peak": round(salu / (cyc * ISSUE_PEAK_SALU), 4),
                                   "frac_of_vector_issue_peak_range": [round(valu / (cyc * ISSUE_PEAK_VALU_PURE), 4), round(valu / (cyc * ISSUE_PEAK_VALU_SFILE), 4)]}
    return out


def sequence_tracks(n_frames=257, tracks=16, w=1280, h=720):
    """sequence tracks through the decoder object the way libheif drives it (one sample per push_data2, pictures polled in output order,
    flush at the end): frames per second of ONE track - every picture is one CABAC critical path, the instance holds one sample at a time - and of
    `tracks` tracks decoded side by side by as many threads (their first pictures and their look-ahead chains coalesce into shared launch sets).  The first pass of each kind checks
    every picture against the CPU oracle."""
    import threading
    import numpy as np
    from oracle import pyoracle as orc
    from libheif_amd.decoder import HipDecoder
    f0 = orc.synth_image(w, h, 8, 1, seed=77)
    frames = [[np.roll(np.roll(p, k // (1 if i == 0 else 2), 0), 2 * k // (1 if i == 0 else 2), 1) for i, p in enumerate(f0)] for k in range(n_frames)]
    kinds = {"lowdelay_ippp_2refs_tmvp_weighted": dict(inter_num_refs=2, temporal_mvp=1, weighted_pred=1),
             "unrestricted_ibbp_tmvp": dict(b_frames=2, inter_num_refs=2, temporal_mvp=1)}
    import libheif_amd
    lib = libheif_amd.load_library()
    lib.hipdec_set_sequence_lookahead.argtypes = [__import__("ctypes").c_int]
    lib.hipdec_set_sequence_lookahead.restype = None
    default_lookahead = int(os.environ.get("HIPDEC_SEQ_LOOKAHEAD", "32"))
    default_pipeline = int(os.environ.get("HIPDEC_SEQ_PIPELINE", "3"))
    from libheif_amd.decoder import set_sequence_pipeline
    res = {"pictures_per_track": n_frames, "size": "%dx%d" % (w, h), "tracks_side_by_side": tracks, "lookahead_samples": default_lookahead, "chains_in_flight": default_pipeline,
           "pipeline": "a chain is enqueued when its look-ahead window is full and its pictures are held back until this many chains are in flight (or the host flushes): "
                       "the CABAC launch of a chain runs beside the pixel steps of the chains in front of it (hipdec_set_sequence_pipeline); *_one_chain_at_a_time = the same "
                       "tracks with every chain waited for where it is launched (rounds 5 / 6 until this change)",
           "lookahead": "behind a track's first picture the decoder gathers this many samples (libheif pushes the next one whenever decode_next_image2 returns no image) "
                        "and decodes them as ONE launch set: one CABAC launch and one motion-derivation launch over all of them, pixel stages in dependency steps "
                        "(hipdec_set_sequence_lookahead); the chains of tracks decoded side by side that ask together share a launch set (hipdec_decoder_chain_stats)"}
    for name, kw in kinds.items():
        # (the VUI names libheif's default nclx, as the stills' streams do: a track without one makes libheif convert every picture on the CPU, context.cc:1533-1543)
        aus = orc.encode_sequence(frames, qp=27, global_mv_x=-8, global_mv_y=-4, inter_skip_pct=30, vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1, **kw)
        if name.startswith("lowdelay"):
            aus_for_libheif = aus
        # (the oracle decodes the first 97 pictures - the track's first picture and three chains, references across chain boundaries included -: its 257
        #  pictures would cost as much CPU time again as the encoder above; whole tracks against the oracle are the GPU tier's business)
        ref = {r["poc"]: r for r in orc.decode_sequence(aus[:97])}

        def play(check, first=None):
            d = HipDecoder()
            got = 0
            try:
                for au in (aus if first is None else aus[:first]) + [None]:
                    if au is not None:
                        d.push_data(au)
                    r = d.next_picture(flush=au is None)
                    while r is not None:
                        if check and got in ref and not all((r[0].planes[c] == ref[got]["planes"][c]).all() for c in range(3)):
                            raise RuntimeError("sequence %s: picture with POC %d differs from the oracle" % (name, got))
                        got += 1
                        r = d.next_picture(flush=au is None)
            finally:
                d.free()
            if got != (len(aus) if first is None else first):
                raise RuntimeError("sequence %s: %d of %d pictures came out" % (name, got, len(aus)))

        play(True)
        t0 = time.perf_counter(); play(False); one = time.perf_counter() - t0
        from libheif_amd.decoder import chain_stats
        before = chain_stats()
        th = [threading.Thread(target=play, args=(False,)) for _ in range(tracks)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        many = time.perf_counter() - t0
        after = chain_stats()
        res[name] = {"one_track_fps": round(n_frames / one, 1), "ms_per_picture": round(one / n_frames * 1e3, 1), "all_tracks_fps": round(tracks * n_frames / many, 1),
                     # side by side: the chains of tracks that ask together run as one launch set (hipdec_decoder_chain_stats)
                     "all_tracks_chains": after[0] - before[0], "all_tracks_launch_sets": after[1] - before[1],
                     "kbytes_per_picture": round(sum(len(a) for a in aus) / len(aus) / 1e3, 1), "verified_against_oracle": True, "verified_pictures": len(ref)}
        set_sequence_pipeline(1)                      # every chain waited for where it is launched
        t0 = time.perf_counter(); play(False); res[name]["one_track_fps_one_chain_at_a_time"] = round(n_frames / (time.perf_counter() - t0), 1)
        th = [threading.Thread(target=play, args=(False,)) for _ in range(tracks)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        res[name]["all_tracks_fps_one_chain_at_a_time"] = round(tracks * n_frames / (time.perf_counter() - t0), 1)
        set_sequence_pipeline(default_pipeline)
        lib.hipdec_set_sequence_lookahead(0)          # round 4's behaviour beside it: every sample decoded at the poll behind its push (the P / B pictures in front of the
        k = 33 if n_frames > 33 else n_frames          # first IRAP of the track's first chunk: one CABAC critical path each)
        t0 = time.perf_counter(); play(False, k); res[name]["one_track_fps_without_lookahead"] = round(k / (time.perf_counter() - t0), 1)
        lib.hipdec_set_sequence_lookahead(default_lookahead)
    # the same kind of track through the REAL libheif (heif_track_decode_next_image: Track_Visual pushes the samples into the plugin and polls it), C threads
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools")) if os.path.join(ROOT, "tools") not in sys.path else None
        import sequence_through_libheif as stl
        r = stl.measure(frames=n_frames, threads_list=(1, tracks), seconds=3.0, pipelines=(default_pipeline,), w=w, h=h, aus=aus_for_libheif)
        res["through_libheif"] = {"workload": r["workload"], "runs": r["runs"]}
    except Exception as ex:   # noqa: BLE001 - reported, does not take the section down
        res["through_libheif"] = {"error": str(ex)[:300]}
    return res


def check_grid_sharded(gs, world):
    """VERDICT round 4 (multi-GPU readiness): a `grid_sharded` section measured with N > 1 must really have used N ranks / N devices - RCCL gather
    with `ranks == N`, peer-copy form with shards on other devices reached by peer access - or it says so loudly instead of reporting a one-rank time
    as an N-GPU time.  Returns (ok, reasons)."""
    if world <= 1:
        return True, []
    why = []
    if "error" in gs:
        why.append("section error: %s" % gs["error"])
    for key in ("wpp", "pps_tiles_4x4"):
        ph = gs.get(key)
        if not isinstance(ph, dict):
            why.append("%s: not measured" % key)
            continue
        r = ph.get("rccl", {})
        if r.get("ranks") != world:
            why.append("%s: RCCL gather ran with ranks=%r, not %d (%s)" % (key, r.get("ranks"), world, r.get("error", "no error reported")))
        elif not r.get("canvas_matches_one_gpu"):
            why.append("%s: the RCCL canvas differs from the one-GPU canvas" % key)
        pc = ph.get("peer_copy")
        if not pc:
            why.append("%s: the peer-copy form was not measured" % key)
        else:
            tr = pc.get("transport", {})
            if tr.get("peer_access_shards", 0) <= 0:
                why.append("%s: no shard reached the root canvas by peer access (%r)" % (key, tr))
            if tr.get("shards_on_root_device", 0) + tr.get("peer_access_shards", 0) + tr.get("runtime_staged_shards", 0) != world:
                why.append("%s: %r shards for %d GPUs" % (key, tr, world))
            if not pc.get("canvas_matches_one_gpu"):
                why.append("%s: the peer-copy canvas differs from the one-GPU canvas" % key)
    return not why, why


_REAL_STDOUT = None


def emit(line):
    """the ONE JSON line, on the process' real stdout (everything else - RCCL's version banner comes through C stdio - went to stderr)"""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)      # C stdio buffers drain into the redirected descriptor, not behind the JSON line
    except Exception:
        pass
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (line + "\n").encode())


def main():
    global _REAL_STDOUT
    a = parse_args()
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)                           # from here on descriptor 1 is stderr
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if a.only_sequences:
        import libheif_amd
        libheif_amd.load_library()
        emit(json.dumps({"sequence_tracks": sequence_tracks()}))
        return
    w, h, def_batch, bit_depth, enc_cfg, out_chroma = WORKLOADS[a.workload]
    enc_cfg = dict(enc_cfg, qp=a.qp)
    for kv in a.enc:
        k, v = kv.split("=")
        enc_cfg[k] = int(v)
    grid = a.workload in ("grid8k", "grid8k_multi")
    grid_single = a.workload == "grid8k"
    extras_on = not (a.no_extras or a.only_main) and world == 1 and not grid
    sharded_on = not (a.no_grid_sharded or a.only_main or grid)
    if grid:
        specs = [(w, h, 2 + t, 8, enc_cfg) for t in range(48)]
        nb = 48
    else:
        nb = a.batch or def_batch
        nd = max(1, min(a.distinct, nb))
        specs = [(w, h, 1 + i, bit_depth, enc_cfg) for i in range(nd)]      # S2 / S4 / S5: seeds 1..nd (every rank cycles the same set,
    distinct = gen_streams(specs, rank, world)                                 # starting at its own offset)
    grid_specs = {}
    if sharded_on or extras_on:
        grid_specs["wpp"] = [(1024, 1024, 2 + t, 8, dict(GRID_VUI, qp=a.qp)) for t in range(48)]
        grid_specs["pps_tiles_4x4"] = [(1024, 1024, 2 + t, 8, dict(GRID_VUI, qp=a.qp, tile_cols=4, tile_rows=4)) for t in range(48)]
    extra_specs = {}
    if extras_on:
        grid_specs["pps_tiles_2x2"] = [(1024, 1024, 2 + t, 8, dict(GRID_VUI, qp=a.qp, tile_cols=2, tile_rows=2)) for t in range(48)]
        extra_specs = {
            "s2_4k_qp17": ("still4k", [(3840, 2160, 1 + i, 8, dict(wpp=1, qp=17)) for i in range(64)], 1024),
            # BASELINE config 5 as written: 1024 x 1080p stills (256 distinct contents)
            "c5_1080p_1024": ("still1080", [(1920, 1080, 1000 + i, 8, dict(wpp=1, qp=a.qp)) for i in range(256)], 1024),
            "s5_1080p_2048": ("still1080", [(1920, 1080, 1000 + i, 8, dict(wpp=1, qp=a.qp)) for i in range(256)], 2048),
            # BASELINE config 4 as written: Main10 4K, YCbCr -> RRGGBB and the PQ -> linear-RGB stage (hipdec_color_pq_to_linear) inside the timed region
            "c4_main10_4k_pq_linear": ("main10_4k", [(3840, 2160, 3 + i, 10, dict(WORKLOADS["main10_4k"][4], qp=a.qp)) for i in range(64)], 512),
            # beyond SURVEY 8(d): the chroma formats added in round 3
            "s6_444_1080p": ("still1080_444", [(1920, 1080, 2000 + i, 8, dict(WORKLOADS["still1080_444"][4], qp=a.qp)) for i in range(32)], 512),
            "s7_422_main10_1080p": ("still1080_422_10", [(1920, 1080, 3000 + i, 10, dict(WORKLOADS["still1080_422_10"][4], qp=a.qp)) for i in range(32)], 512),
        }
        extra_streams = {k: gen_streams(v[1]) for k, v in extra_specs.items()}
    grid_streams = {k: gen_streams(v, rank, world) for k, v in grid_specs.items()}
    # BASELINE config 5 as written at N > 1: 1024 x 1080p stills interleaved over the GPUs (still i -> GPU i mod N; at N = 1 it is extra_workloads.c5_1080p_1024)
    c5_on = (world > 1 or os.environ.get("HIPDEC_BENCH_C5")) and not (a.no_extras or a.only_main or grid) and a.workload == "still4k"   # (env: exercise the section on one GPU)
    c5_streams = gen_streams([(1920, 1080, 1000 + i, 8, dict(wpp=1, qp=a.qp)) for i in range(256)], rank, world) if c5_on else None

    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decode path has no CPU fallback")
    torch.cuda.set_device(local_rank)

    import ctypes as C
    import numpy as np
    import libheif_amd
    from libheif_amd.decoder import Batch, HipDecoder
    from libheif_amd._capi import check, DeviceBuffer
    from libheif_amd.grid import GridDecoderC, GridDecoderRccl, GridLayout, RcclComm

    lib = libheif_amd.load_library()
    check(lib.hipdec_init(local_rank))
    props = torch.cuda.get_device_properties(local_rank)
    cu_count = int(props.multi_processor_count)
    clock_ghz = float(getattr(props, "clock_rate", 2400000)) / 1e6

    def sync():
        check(lib.hipdec_stream_synchronize(None))
        torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed(step_fn, steps, warmup, before_timed=None):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; max over ranks"""
        for _ in range(warmup):
            step_fn()
        sync()
        if before_timed:
            before_timed()
        barrier(); sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        sync(); barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed

    def grid_run(streams48, devices, steps=2, warmup=1, want_hash=False):
        """one 8K grid photo through hipdec_grid_* on `devices` (this process drives them all): ms per photo, transport, canvas hash"""
        g = GridDecoderC({t: streams48[t] for t in range(48)}, GridLayout(6, 8, 1024, 1024, 8192, 6144), devices)
        rgb = torch.empty((6144, 8192 * 3), dtype=torch.uint8, device="cuda:%d" % devices[0])

        def st():
            g.decode(); g.to_rgb(10, out_dev=(rgb.data_ptr(), rgb.stride(0)))
        for _ in range(warmup):
            st()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            st()
        sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        n = [C.c_int(), C.c_int(), C.c_int()]
        lib.hipdec_grid_transport.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
        check(lib.hipdec_grid_transport(g._h, C.byref(n[0]), C.byref(n[1]), C.byref(n[2])))
        hsh = None
        if want_hash:
            import hashlib
            hsh = hashlib.sha1(rgb.cpu().numpy().tobytes()).hexdigest()
        g.free()
        return ms, {"shards_on_root_device": n[0].value, "peer_access_shards": n[1].value, "runtime_staged_shards": n[2].value}, hsh

    # ------------------------------------------------------------------------------------------------------------------
    # main measurement, resident form: inputs in HBM when the timed region starts
    # ------------------------------------------------------------------------------------------------------------------
    gd = None
    wl = None
    rotated = distinct
    if grid:
        # the product path: hipdec_grid_* in C++ (include/heif_hipdec.h).  grid8k: ONE photo, rank 0 shards its 48 tiles t mod N over devices
        # 0..N-1, every decoded tile is pasted into the canvas on device 0 by a strided peer copy (xGMI), the colour stage runs over the
        # canvas; the other ranks only take part in the barriers (strong scaling).  grid8k_multi: every rank decodes its own photos on its
        # own GPU (photo i -> GPU i mod N; weak scaling).
        layout = GridLayout(6, 8, w, h, 8 * w, 6 * h)
        n_items, px_rank, bs_bytes = 48, w * h * 48, sum(len(x) for x in distinct)
        total_px = w * h * 48 * (1 if grid_single else world)
        batch = None
        if grid_single:
            if rank == 0:
                gd = GridDecoderC({t: distinct[t] for t in range(48)}, layout, list(range(world)))
                rgb_out = torch.empty((6 * h, 8 * w * 3), dtype=torch.uint8, device="cuda:0")
        else:
            gd = GridDecoderC({t: distinct[t] for t in range(48)}, layout, [local_rank])
            rgb_out = torch.empty((6 * h, 8 * w * 3), dtype=torch.uint8, device="cuda:%d" % local_rank)

        def step():
            if gd is not None:
                gd.decode()
                gd.to_rgb(10, out_dev=(rgb_out.data_ptr(), rgb_out.stride(0)))     # waits for the shards, colour stage into HBM
    else:
        off = (rank * len(distinct)) // max(1, world)
        lib.hipdec_set_stage_overlap(1 if a.parts > 1 else 0)
        rotated = distinct[off:] + distinct[:off]
        wl = Workload(lib, a.workload, rotated, nb, out_chroma, w, h, bit_depth, a.parts)
        wl.make_resident()
        batch = wl.batch
        n_items, px_rank, bs_bytes = wl.n, wl.px, wl.bs_bytes
        total_px = wl.px * world
        step = wl.step_resident
    if batch is not None:
        wl.timing_slots(max(1, a.steps))
    elapsed = timed(step, a.steps, a.warmup, before_timed=(lambda: (wl.status() if a.warmup else None, wl.timing_slots(max(1, a.steps)))) if batch is not None else None)
    if batch is not None:
        wl.status()            # device-side decode errors are loud
    elif gd is not None:
        gd.wait()
    ms_resident = elapsed / a.steps * 1e3
    value_resident = total_px / (elapsed / a.steps) / 1e6
    avg_us = kernel_times(wl.batches, a.steps) if not grid else None

    # config 5 over the job's GPUs: every rank decodes ITS stills (i mod N == rank) of the 1024 as one batch, no data-path collective; barrier + max over
    # ranks as the main measurement.  Total work is fixed: the line says "strong".  (A rank that cannot build its batch takes the others out of the
    # section before any of them enters a barrier.)
    c5_line = None
    if c5_on:
        c5w, c5_err = None, ""
        try:
            mine = [c5_streams[i % 256] for i in range(1024) if i % world == rank]
            c5w = Workload(lib, "still1080", mine, len(mine), 10, 1920, 1080, 8, 1)
            c5w.make_resident()
        except Exception as ex:   # noqa
            c5w, c5_err = None, str(ex)[:200]
        flag = torch.tensor([1 if c5w is not None else 0], dtype=torch.int32, device="cuda")
        if dist is not None:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            el5 = timed(c5w.step_resident, 2, 1)
            c5w.status()
            c5_line = {"workload": "1024 x 1920x1080 HEIC 4:2:0 8-bit stills, still i -> GPU i mod %d (%d per GPU), fused YCbCr->RGB24, inputs resident" % (world, len(mine)),
                       "value": round(1024 * 1920 * 1080 / (el5 / 2) / 1e6, 2), "unit": "Mpixel/s", "ms_per_step": round(el5 / 2 * 1e3, 3), "n_gpus": world,
                       "scaling": "strong", "one_gpu": "extra_workloads.c5_1080p_1024 of the N = 1 line"}
        else:
            c5_line = {"error": "a rank could not build its batch: " + (c5_err or "another rank")}
        if c5w is not None:
            c5w.free()

    # what the timed steps produced for stills spread over the batch (checked against the CPU oracle further down; outside every timed region)
    check_items = {}
    verify_on = batch is not None and rank == 0 and not a.only_main and not a.no_cpu_baseline and a.parts == 1 and world == 1
    if verify_on:
        import hashlib
        nv = max(1, min(a.verify_stills, n_items))
        picks = sorted(set(min(n_items - 1, k * (n_items // nv) + k) for k in range(nv)))
        for i in picks:
            got_planes = batch.planes(i)
            got_rgb = batch.rgb(i)
            check_items[i] = {"content": i % len(rotated),
                              "planes": [hashlib.sha1(np.ascontiguousarray(p, dtype=np.uint16).tobytes()).hexdigest() for p in got_planes],
                              "rgb": hashlib.sha1(np.ascontiguousarray(got_rgb).tobytes()).hexdigest()}

    out = None
    if rank == 0:
        s = 2 if bit_depth > 8 else 1
        s_out = 2 if out_chroma in (12, 14) else 1
        beta = bs_bytes / px_rank
        out = {
            "metric": "Mpixels/s HEIC 4:2:0 8-bit decode" if bit_depth == 8 else "Mpixels/s HEIC 4:2:0 %d-bit decode" % bit_depth,
            "value": round(value_resident, 2), "unit": "Mpixel/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_resident, 3),
            "higher_is_better": True, "scaling": "strong" if grid_single and grid else "weak", "vs_baseline": None,
            "dtype": "u8" if bit_depth == 8 else "u16",
            "native_library": os.path.relpath(__import__("libheif_amd").library_path(), ROOT),   # (which build produced the numbers)
            "data": "synthetic (seeded noise+gradient stills coded by the test-only HEVC intra encoder, QP %d, %d distinct contents)" % (a.qp, len(distinct)),
            "config": {"workload": (("one 8192x6144 grid photo = 48 tiles of 1024x1024, tiles t mod N over N GPUs in one process (hipdec_grid_*), peer-copy paste + RGB24 on GPU 0"
                                     if grid_single else "one 8192x6144 grid photo (48 tiles of 1024x1024) per GPU and step through hipdec_grid_* + RGB24; photo i -> GPU i mod N") if grid else
                                    "%d x %dx%d HEIC 4:2:0 %d-bit stills per GPU and step, WPP, CTB 64, fused YCbCr->%s" %
                                    (n_items, w, h, bit_depth, "RGB24" if out_chroma == 10 else "RRGGBB")),
                       "timed_region": "inputs resident in HBM: hipdec_batch_run_rgb (decode + colour stage) per step",
                       "batches_per_step": a.parts if not grid else 1,
                       "stage_overlap": bool(a.parts > 1 and not grid),
                       "stills_per_step_per_gpu": n_items, "distinct_contents": len(distinct), "bitstream_bytes_per_px": round(beta, 4),
                       "substreams_per_still": batch.info(0)["num_substreams"] if batch is not None else 16,
                       "parallelism": (("tiles sharded over %d GPUs, one process" % world) if grid_single else "independent photos, replicas x%d" % world) if grid else "replicas x%d" % world},
            "value_resident": round(value_resident, 2),
            "resident": {"value": round(value_resident, 2), "unit": "Mpixel/s", "ms_per_step": round(ms_resident, 3),
                         "timed_region": "inputs resident in HBM when the timed region starts: hipdec_batch_run_rgb per step"},
        }
        if c5_line is not None:
            out["config5_interleaved"] = c5_line
        if avg_us is not None:
            coded = coded_fraction(batch)
            alg = alg_bytes(beta, coded, s, s_out)
            kernels = kernel_table(avg_us, alg, px_rank)
            dom = max(kernels, key=lambda k: kernels[k]["avg_us"])
            # HBM traffic of the dominant kernel: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE separately) over this same
            # command line, recorded per luma pixel in profiles/pmc_traffic.json by tools/prof_hbm_traffic.sh; only used when
            # the recorded workload is the one benchmarked now
            traffic, traffic_source = None, None
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                if rec.get("stills_per_step") == n_items and rec.get("workload") == a.workload and rec.get("qp") == a.qp:
                    traffic = round(rec["bytes_per_px"][KERNEL_NAMES[dom]] * px_rank, 0)
                    traffic_source = "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command line, kernels of commit %s" % rec.get("commit", "?")
            except Exception:
                pass
            # `bound` names the roofline achieved / peak are priced on (the contract's "hbm" | "mfma"); `limited_by` names what the dominant kernel is
            # really bound by - instruction issue for the CABAC parser and the reconstruction wavefront (never label them HBM-bound)
            out["roofline"] = dict(bound="hbm", limited_by=("issue" if dom in ("parse", "recon") else "hbm"), kernel=KERNEL_NAMES[dom],
                                   achieved=kernels[dom]["achieved_gbs"], peak=HBM_PEAK_GBS, unit="GB/s",
                                   frac=kernels[dom]["frac"], traffic=traffic, traffic_source=traffic_source,
                                   note="dominant kernel by device time, priced in algorithmic HBM bytes as the contract asks; CABAC parsing is bound by "
                                        "instruction issue (one dependency chain per substream), not by HBM: its real yardstick is `issue_roofline` "
                                        "(DESIGN.md section 4); streaming kernels: see `kernels`")
            out["kernels"] = kernels
            ir = issue_roofline(kernels, px_rank, cu_count, clock_ghz)
            if ir:
                out["issue_roofline"] = ir
            out["coded_samples_per_px"] = round(coded, 4)
            e2e = (beta + 3.0 * s + 3.0 * s_out) * px_rank    # drop-in end-to-end bytes: beta + 1.5 s + 1.5 s + 3 s_out
            out["end_to_end"] = {"alg_bytes_per_px": round(beta + 3.0 * s + 3.0 * s_out, 3), "achieved_gbs": round(e2e / (elapsed / a.steps) / 1e9, 2),
                                 "frac_of_hbm_peak": round(e2e / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS, 5)}

    # ------------------------------------------------------------------------------------------------------------------
    # the same workload from compressed bytes in host memory: batch_create inside the step, double-buffered.
    # This region is the headline `value` (round 4).
    # ------------------------------------------------------------------------------------------------------------------
    if not grid and not a.only_main:
        # one chain of batches per part: batch k+1 of a chain takes over batch k's arena and RGB buffers
        chains = []
        state = {"host_s": 0.0, "creates": 0}

        def create(streams, recycle, rgb_state):
            t = time.perf_counter()
            b = Batch(streams, recycle=recycle)   # host parse (worker threads) + pinned staging + asynchronous upload into the predecessor's
            b.use_rgb(rgb_state)                  # arena, ordered behind the predecessor's kernels
            state["host_s"] += time.perf_counter() - t
            state["creates"] += 1
            return b

        for j, b0 in enumerate(wl.batches):
            rs = b0.rgb_state()
            chains.append({"streams": wl.part_streams[j], "rgb": rs, "prev": b0, "next": create(wl.part_streams[j], b0, rs)})
        wl.batches = []

        def step_host():
            for ch in chains:
                cur = ch["next"]
                cur.run_rgb()
                ch["next"] = create(ch["streams"], cur, ch["rgb"])   # host work for the chain's next batch overlaps the kernels in flight
                ch["prev"].status()             # (arena already handed on: the status word was copied back behind its kernels)
                ch["prev"].free()
                ch["prev"] = cur

        def reset_counters():
            state["host_s"], state["creates"] = 0.0, 0

        el_h = timed(step_host, a.steps, a.warmup, before_timed=reset_counters)
        for ch in chains:
            ch["prev"].status()
            ch["prev"].free()
            ch["next"].free()
        if rank == 0:
            v_h = total_px / (el_h / a.steps) / 1e6
            out["value"] = round(v_h, 2)                    # SURVEY 8(d)'s wall-clock definition is the headline
            out["ms_per_step"] = round(el_h / a.steps * 1e3, 3)
            out["value_from_host_bytes"] = round(v_h, 2)    # (round-3 key, kept)
            out["config"]["timed_region"] = ("compressed bytes in host memory -> planes + RGB complete in HBM (SURVEY 8d): hipdec_batch_create_recycling "
                                             "(header parsing, pinned staging, asynchronous H2D upload into the predecessor's arena) + run + colour per step; "
                                             "the host work of batch k+1 overlaps the kernels of batch k.  `value_resident`: inputs already in HBM")
            out["from_host_bytes"] = {
                "value": round(v_h, 2), "unit": "Mpixel/s", "ms_per_step": round(el_h / a.steps * 1e3, 3),
                "host_ms_per_batch_create": round(state["host_s"] / max(1, state["creates"]) * 1e3, 2), "batches_per_step": len(wl.part_streams),
                "h2d_bytes_per_step": wl.bs_bytes}
        del chains

    # ------------------------------------------------------------------------------------------------------------------
    # batch-size curve: stills per batch -> throughput, resident form (the parser's t(N) = a + b / N)
    # ------------------------------------------------------------------------------------------------------------------
    if extras_on and rank == 0 and a.workload == "still4k":
        curve = []
        for n in (64, 128, 256, 512, 1024):
            if n >= nb:
                continue
            e = Workload(lib, a.workload, rotated, n, out_chroma, w, h, bit_depth, 1)
            e.make_resident()
            e.timing_slots(2)
            el = timed(e.step_resident, 2, 1, before_timed=lambda: (e.status(), e.timing_slots(2)))
            e.status()
            t = kernel_times(e.batches, 2)
            curve.append({"stills": n, "value_resident": round(e.px / (el / 2) / 1e6, 1), "ms_per_step": round(el / 2 * 1e3, 2), "parse_ms": round(t["parse"] / 1e3, 2)})
            e.free()
        curve.append({"stills": nb, "value_resident": round(value_resident, 1), "ms_per_step": round(ms_resident, 2), "parse_ms": round(avg_us["parse"] / 1e3, 2)})
        out["batch_curve"] = curve

    # ------------------------------------------------------------------------------------------------------------------
    # single-still latency form (BASELINE config 2 as written), plugin life cycle
    # ------------------------------------------------------------------------------------------------------------------
    if rank == 0 and not a.only_main and not grid:
        lib.hipdec_set_stage_overlap(0)
        first = distinct[0]
        single = Batch([first])
        single.alloc_rgb(out_chroma)
        for _ in range(2):
            single.run(); single.to_rgb_all()
        sync()
        ts = time.perf_counter()
        reps = 5
        for _ in range(reps):
            single.run(); single.to_rgb_all()
        sync()
        single_ms = (time.perf_counter() - ts) / reps * 1e3
        single_t = single.kernel_timing_us()
        # the plugin life cycle on one still, host to host (new_decoder -> push_data -> decode -> D2H of the planes -> free):
        # what heif_decode_image() pays per item through libheif, PCIe included
        tp = time.perf_counter()
        for _ in range(3):
            dec = HipDecoder(); dec.push_data(first); dec.decode_next_image(); dec.free()
        plugin_ms = (time.perf_counter() - tp) / 3 * 1e3
        single.free()
        out["single_still"] = {"ms": round(single_ms, 3), "mpixel_s": round(w * h / single_ms / 1e3, 2),
                               "kernel_us": {k: round(v, 1) for k, v in single_t.items()},
                               "plugin_lifecycle_host_to_host_ms": round(plugin_ms, 3)}

    # ------------------------------------------------------------------------------------------------------------------
    # the other synthetic inputs of SURVEY 8(d) / BASELINE's configs as written, resident form, 1 warm-up + 2 timed steps each
    # ------------------------------------------------------------------------------------------------------------------
    if extras_on and rank == 0:
        lib.hipdec_set_stage_overlap(1 if a.parts > 1 else 0)
        extras = {}
        for key in ("wpp", "pps_tiles_2x2", "pps_tiles_4x4"):
            ms, _, _ = grid_run(grid_streams[key], [0])
            nsub = {"wpp": 16, "pps_tiles_2x2": 32, "pps_tiles_4x4": 64}[key]
            px = 1024 * 1024 * 48
            extras["s3_grid8k_" + key] = {"workload": "one 8192x6144 grid photo = 48 tiles of 1024x1024 (%d substreams per tile, %d in all), hipdec_grid_* + RGB24 on one GPU" % (nsub, 48 * nsub),
                                          "value": round(px / ms / 1e3, 2), "unit": "Mpixel/s", "ms_per_step": round(ms, 3),
                                          "bitstream_bytes_per_px": round(sum(len(x) for x in grid_streams[key]) / px, 4)}
        lib.hipdec_color_pq_to_linear.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        for key, (wname, sp, n) in extra_specs.items():
            ew, eh, _, ebd, _, eout = WORKLOADS[wname]
            st = extra_streams[key]
            e = Workload(lib, wname, st, n, eout, ew, eh, ebd, a.parts)
            e.make_resident()
            eb = e.batch
            step_fn = e.step_resident
            pq = "pq_linear" in key
            lin = None
            if pq:
                # config 4: PQ code values of the RRGGBB rows -> linear light, float32 R, G, B per pixel (12 B/px written), one launch per still on the
                # library's stream right behind the batch's colour stage
                lin = [DeviceBuffer(ew * eh * 12) for _ in range(n)]
                rgb_bufs = [buf.ptr for buf, _, _ in eb._rgb]

                def step_fn(e=e, lin=lin, rgb_bufs=rgb_bufs, ew=ew, eh=eh, ebd=ebd):
                    e.step_resident()
                    for i in range(len(lin)):
                        check(lib.hipdec_color_pq_to_linear(rgb_bufs[i], ew * 6, ew, eh, 3, ebd, 0, lin[i].ptr, ew * 12, None))
            e.timing_slots(2)
            el = timed(step_fn, 2, 1, before_timed=lambda: (e.status(), e.timing_slots(2)))
            e.status()
            eavg = kernel_times(e.batches, 2)
            es, eso = (2 if ebd > 8 else 1), (2 if eout in (12, 14) else 1)
            ebeta = e.bs_bytes / e.px
            ecf = sp[0][4].get("chroma_format_idc", 1)
            espp, ecw = {1: (1.5, 0.25), 2: (2.0, 0.5), 3: (3.0, 1.0)}[ecf]
            ekt = kernel_table(eavg, alg_bytes(ebeta, coded_fraction(eb, ecw), es, eso, espp), e.px, colour_stage=eout is not None)
            extras[key] = {"workload": "%d x %dx%d %d-bit %s stills (%d distinct), QP %d, %s" %
                                       (n, ew, eh, ebd, {1: "4:2:0", 2: "4:2:2", 3: "4:4:4"}[ecf], len(st), sp[0][4].get("qp", a.qp),
                                        "planes only" if eout is None else ("YCbCr->" + ("RGB24" if eout == 10 else "RRGGBB") + (" fused into SAO" if ecf == 1 and eout == 10 else "") +
                                                                            (" + PQ->linear float32 RGB (hipdec_color_pq_to_linear) in the timed region" if pq else ""))),
                           "value": round(e.px / (el / 2) / 1e6, 2), "unit": "Mpixel/s", "ms_per_step": round(el / 2 * 1e3, 3),
                           "bitstream_bytes_per_px": round(ebeta, 4),
                           "kernels": ekt}
            if pq:
                extras[key]["pq_stage_ms_per_step"] = round(el / 2 * 1e3 - eavg["decode_total"] / 1e3 - eavg["colour"] / 1e3, 3)
                lin = None
            e.free()
        out["extra_workloads"] = extras
        try:
            out["sequence_tracks"] = sequence_tracks()
        except Exception as ex:   # noqa: a failure here is reported, it does not take the line down
            out["sequence_tracks"] = {"error": str(ex)[:300]}
        # BASELINE.json's configs as written (config 1 is the CPU plumbing case)
        ss = out.get("single_still", {})
        out["baseline_configs"] = {
            "config2_single_4k_still_fused_rgb": {"ms": ss.get("ms"), "mpixel_s": ss.get("mpixel_s")},
            "config3_8k_grid_48_tiles_one_gpu": {"ms": extras["s3_grid8k_wpp"]["ms_per_step"], "mpixel_s": extras["s3_grid8k_wpp"]["value"],
                                                 "sharded_over_n_gpus": "see grid_sharded"},
            "config4_main10_4k_pq_to_linear_rgb": {"stills": 512, "mpixel_s": extras["c4_main10_4k_pq_linear"]["value"], "ms_per_step": extras["c4_main10_4k_pq_linear"]["ms_per_step"]},
            "config5_1024_x_1080p": {"mpixel_s": extras["c5_1080p_1024"]["value"], "ms_per_step": extras["c5_1080p_1024"]["ms_per_step"]},
        }

    # ------------------------------------------------------------------------------------------------------------------
    # grid_sharded: ONE 8K grid photo (BASELINE config 3) over the N GPUs of the job: peer copies (one process) and RCCL (SPMD)
    # ------------------------------------------------------------------------------------------------------------------
    if sharded_on:
        res = {}

        def section():
            photos = {}
            for key in ("wpp", "pps_tiles_4x4"):
                st = grid_streams[key]
                ph = {}
                ref_hash = None
                if rank == 0:
                    one_ms, _, ref_hash = grid_run(st, [0], want_hash=True)
                    ph["one_gpu_ms"] = round(one_ms, 3)
                    ph["one_gpu_mpixel_s"] = round(8192 * 6144 / one_ms / 1e3, 1)
                    if world > 1:
                        pm, transport, ph_hash = grid_run(st, list(range(world)), want_hash=True)
                        ph["peer_copy"] = {"ms": round(pm, 3), "mpixel_s": round(8192 * 6144 / pm / 1e3, 1), "transport": transport, "canvas_matches_one_gpu": ph_hash == ref_hash,
                                           "strong_scaling_efficiency": round(one_ms / (world * pm), 4)}
                barrier()
                # SPMD: every rank decodes its tiles, grouped ncclSend / ncclRecv to rank 0 inside libheifhip.so
                if lib.hipdec_rccl_available():
                    def exchange(raw):
                        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
                        dist.broadcast(t, 0)
                        return bytes(t.cpu().numpy().tobytes())
                    comm = RcclComm(rank, world, exchange if world > 1 else None)
                    g = GridDecoderRccl({t: st[t] for t in range(48)}, GridLayout(6, 8, 1024, 1024, 8192, 6144), comm)
                    rgb = torch.empty((6144, 8192 * 3), dtype=torch.uint8, device="cuda") if rank == 0 else None

                    def rstep():
                        g.decode()
                        if rank == 0:
                            g.to_rgb(10, out_dev=(rgb.data_ptr(), rgb.stride(0)))
                        else:
                            g.wait()
                    el = timed(rstep, 2, 1)
                    if rank == 0:
                        import hashlib
                        rm = el / 2 * 1e3
                        ph["rccl"] = {"ms": round(rm, 3), "mpixel_s": round(8192 * 6144 / rm / 1e3, 1), "ranks": world,
                                      "canvas_matches_one_gpu": hashlib.sha1(rgb.cpu().numpy().tobytes()).hexdigest() == ref_hash,
                                      "strong_scaling_efficiency": round(ph["one_gpu_ms"] / (world * rm), 4)}
                    g.free(); comm.free()
                elif rank == 0:
                    ph["rccl"] = {"error": "librccl not loadable"}
                photos[key] = ph
            res["photos"] = photos

        import threading

        def guarded_section():
            try:
                torch.cuda.set_device(local_rank)
                section()
            except BaseException as e:      # reported in the line, never fatal for the main measurement
                res["error"] = repr(e)[:400]
        th = threading.Thread(target=guarded_section, daemon=True)
        t_sec = time.perf_counter()
        th.start()
        th.join(timeout=420.0)
        hung = th.is_alive()
        if rank == 0:
            gs = {"photo": "8192x6144 = 48 tiles of 1024x1024 (BASELINE config 3), tile t -> GPU t mod N, canvas + RGB24 on GPU 0; `wpp`: 16 CABAC substreams per tile, "
                           "`pps_tiles_4x4`: 64 (SURVEY 8e: one photo scales with the substreams inside a tile, not with GPUs)",
                  "n_gpus": world, "seconds": round(time.perf_counter() - t_sec, 1)}
            gs.update(res.get("photos", {}))
            if "error" in res:
                gs["error"] = res["error"]
            if hung:
                gs["error"] = "section did not finish within 420 s (abandoned; the main measurement above is unaffected)"
            ok, why = check_grid_sharded(gs, world)
            gs["multi_gpu_check"] = {"ok": ok, "requires": "rccl.ranks == n_gpus, peer_access_shards > 0, canvases equal to the one-GPU canvas", "failed": why}
            if not ok:
                sys.stderr.write("bench.py: grid_sharded DID NOT RUN ACROSS %d GPUs AS CLAIMED: %s\n" % (world, "; ".join(why)))
            out["grid_sharded"] = gs
        if hung or (world > 1 and "error" in res):
            # a rank that failed or hangs inside a collective cannot be waited for: the line goes out with what was measured
            if rank == 0:
                emit(json.dumps(out))
            os._exit(0)

    # ------------------------------------------------------------------------------------------------------------------
    # the drop-in form: T application threads x heif_decode_image() on distinct 4K HEIC files through the UNMODIFIED reference libheif
    # (oracle/_ref/libheif.so, the host application here, prebuilt) + the plugin, host to host (tools/dropin_throughput.py, C pthreads)
    # ------------------------------------------------------------------------------------------------------------------
    if rank == 0 and not a.only_main and not a.no_dropin and not a.no_extras and world == 1 and not grid and a.workload == "still4k":
        try:
            from tools import dropin_throughput
            # the thread counts the reference itself runs at (4 decoding threads per context, libheif/context.h:72; tests/test-race.go: 100) beside the
            # counts the coalescer needs for throughput; per call: mean and p95 wall time of heif_decode_image()
            d = dropin_throughput.measure((1, 8, 32, 64, 256, 1024), n_files=64, seconds=3.0, qp=a.qp)
            try:   # ... and to interleaved RGB24 through the patched libheif (HIP colour op, libheif_amd/integration) where oracle/_ref holds one
                if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libheif_hipcolor.so")):
                    d["rgb24_through_patched_libheif"] = dropin_throughput.measure((64, 256), n_files=64, seconds=3.0, qp=a.qp, rgb=True, libheif="libheif_hipcolor.so")["runs"]
            except Exception as e:
                d["rgb24_through_patched_libheif"] = {"error": repr(e)[:300]}
            out["dropin_through_libheif"] = d
        except Exception as e:   # the reference build is test infrastructure: its absence must not fail the bench
            out["dropin_through_libheif"] = {"error": repr(e)[:300]}

    if rank == 0 and not a.no_cpu_baseline and not a.only_main and world == 1 and not grid:
        contents = sorted(set(v["content"] for v in check_items.values())) or [0]
        out["cpu_baseline"] = cpu_baseline([rotated[c] for c in contents], w * h, a.cpu_seconds, a.cpu_procs, bit_depth)
        ref = dict(zip(contents, out["cpu_baseline"].pop("_hashes")))
        if check_items:
            bad = [i for i, v in check_items.items() if ref.get(v["content"]) is None or v["planes"] != ref[v["content"]]["planes"] or v["rgb"] != ref[v["content"]]["rgb"]]
            bad_planes = [i for i in bad if ref.get(check_items[i]["content"]) is None or check_items[i]["planes"] != ref[check_items[i]["content"]]["planes"]]
            out["verified"] = {"stills": sorted(check_items), "count": len(check_items), "of": n_items,
                               "against": "CPU oracle decode + the reference's colour op of each still's stream (sha1 of Y, Cb, Cr and of the RGB rows)",
                               "planes_match": not bad_planes, "rgb_match": not bad, "mismatching_stills": bad}
            if bad:
                emit(json.dumps(out))
                raise SystemExit("bench.py: decoded stills %r differ from the CPU oracle" % bad)
    barrier()
    if rank == 0:
        emit(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def _cpu_worker(arg):
    stream, budget_s, max_n, bit_depth, want_hash = arg
    from oracle import pyoracle as orc
    n = 0
    hashes = None
    t0 = time.perf_counter()
    while True:
        r = orc.decode(stream)
        y, cb, cr = r["planes"]
        if bit_depth == 8:
            nclx = tuple(r["nclx"])
            m = 6 if nclx[2] == 2 else nclx[2]
            if nclx[3] and m not in (0, 8):      # the reference planner's rule: integer op for full range, else the float chain
                rgb = orc.color_420_to_rgb24(y, cb, cr, nclx)
            else:
                rgb = orc.color_rgb_planar_to_interleaved8(*orc.color_ycbcr_to_rgb_planar(y, cb, cr, 8, 1, nclx))
        else:
            rgb = orc.color_420_to_rrggbb(y, cb, cr, bit_depth, tuple(r["nclx"]))
        n += 1
        if want_hash and hashes is None:
            import hashlib
            import numpy as np
            hashes = {"planes": [hashlib.sha1(np.ascontiguousarray(p, dtype=np.uint16).tobytes()).hexdigest() for p in (y, cb, cr)],
                      "rgb": hashlib.sha1(np.ascontiguousarray(rgb).tobytes()).hexdigest()}
        if time.perf_counter() - t0 > budget_s or n >= max_n:
            break
    return n, time.perf_counter() - t0, hashes


def cpu_baseline(streams, px, budget_s, procs, bit_depth=8):
    """The CPU oracle (a scalar, spec-literal port: libde265 itself is not available here) on this host: one job per still of `streams`
    (distinct stills of the bench workload: the same decodes give the hashes the GPU results are checked against), at least `procs` jobs
    on `procs` processes, each decoding its still (+ the reference's 4:2:0->RGB ops) for a bounded time; throughput = all decodes / wall."""
    import multiprocessing as mp
    procs = procs or min(32, os.cpu_count() or 1)
    n_jobs = max(procs, len(streams))
    per_proc_s = max(1.0, budget_s / 2)      # ~2 x budget_s core-seconds per process pair keeps the run short
    jobs = [(streams[i % len(streams)], per_proc_s, 6, bit_depth, i < len(streams)) for i in range(n_jobs)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_cpu_worker, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    n = sum(r[0] for r in res)
    return {"_hashes": [res[i][2] for i in range(len(streams))], "value": round(px * n / wall / 1e6, 2), "unit": "Mpixel/s", "cores": procs, "kind": "port",
            "sample": "%d processes, %d jobs x ~%d decode(s) of %d distinct stills of the bench workload (CPU oracle decode + the reference's "
                      "4:2:0->RGB ops), %.1f s wall, %.0f core-seconds" % (procs, n_jobs, max(1, n // n_jobs), len(streams), wall, sum(r[1] for r in res))}


if __name__ == "__main__":
    main()
