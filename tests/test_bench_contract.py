"""The bench line contract (one JSON object with the driver's fields plus `roofline` and `cpu_baseline`), checked on the line
bench.py printed on the MI355X at the end of the round (committed under profiles/ as r<NN>_final_bench.json)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_every_contract_field():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_bench.json")))
    assert files, "no bench line committed under profiles/"
    text = open(files[-1]).read().strip()
    assert len(text.splitlines()) == 1, "bench.py prints ONE JSON line"
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "Mpixel/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u8" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert abs(d["value"] - d["config"]["stills_per_step_per_gpu"] * 3840 * 2160 * d["n_gpus"] / (d["ms_per_step"] * 1e3)) / d["value"] < 1e-3
    # round 3: the run checks a result against the CPU oracle and reports the SURVEY 8(d) from-host form as a first-class key
    assert d["verified"]["planes_match"] is True and d["verified"]["rgb_match"] is True
    assert d["value_from_host_bytes"] > 0 and d["single_still"]["ms"] > 0
    if os.path.basename(files[-1]) >= "r04":
        # round 4: `value` is the from-host-bytes number of SURVEY 8(d) (the resident form beside it), >= 32 stills spread over the batch are verified,
        # the scalar-issue roofline of the CABAC kernel stands beside the HBM one, BASELINE's other configurations and the batch curve are in the line
        assert d["value_resident"] >= d["value"] > 0 and abs(d["value"] - d["value_from_host_bytes"]) / d["value"] < 1e-6
        assert d["verified"]["count"] >= 32 and len(set(d["verified"]["stills"])) == d["verified"]["count"] and d["verified"]["mismatching_stills"] == []
        assert max(d["verified"]["stills"]) - min(d["verified"]["stills"]) > d["verified"]["of"] // 2
        assert 0 < d["issue_roofline"]["kernels"]["parse"]["frac_of_scalar_issue_peak"] <= 1.0
        assert [p["stills"] for p in d["batch_curve"]] == sorted(p["stills"] for p in d["batch_curve"]) and len(d["batch_curve"]) >= 5
        for k in ("config2_single_4k_still_fused_rgb", "config3_8k_grid_48_tiles_one_gpu", "config4_main10_4k_pq_to_linear_rgb", "config5_1024_x_1080p"):
            assert d["baseline_configs"][k]["mpixel_s"] > 0, k
        g = d["grid_sharded"]
        assert g["wpp"]["one_gpu_ms"] > 0 and g["wpp"]["rccl"]["canvas_matches_one_gpu"] is True and g["wpp"]["rccl"]["ranks"] == d["n_gpus"]
        if "sequence_tracks" in d:     # SURVEY 8 f3 measured: P and B tracks through the decoder object, checked against the oracle inside the run
            st = d["sequence_tracks"]
            assert "error" not in st, st
            for k in ("lowdelay_ippp_2refs_tmvp_weighted", "unrestricted_ibbp_tmvp"):
                assert st[k]["one_track_fps"] > 0 and st[k]["all_tracks_fps"] > st[k]["one_track_fps"] and st[k]["verified_against_oracle"] is True


def test_issue_roofline_covers_every_kernel_of_the_pmc_record():
    """bench.py's issue_roofline applies profiles/pmc_issue.json (instructions per luma pixel, one PMC pass) to the run's kernel times: every kernel
    the record holds gets its share of the scalar / vector issue peaks, and none may exceed a peak"""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_issue.json")))
    assert {"k_parse", "k_recon", "k_residual", "k_sao", "k_deblock"} <= set(rec["insts_per_px"])
    kern = {"parse": {"kernel": "k_parse", "avg_us": 738604.4}, "recon": {"kernel": "k_recon", "avg_us": 128503.6}, "residual": {"kernel": "k_residual", "avg_us": 83848.1},
            "deblock": {"kernel": "k_deblock", "avg_us": 21106.3}, "sao_rgb": {"kernel": "k_sao", "avg_us": 53736.4}}
    r = bench.issue_roofline(kern, 2048 * 3840 * 2160, 256, 2.4)
    assert set(r["kernels"]) == set(kern)
    for k, v in r["kernels"].items():
        lo, hi = v["frac_of_vector_issue_peak_range"]   # [every VALU instruction VGPR-only, every one touching the scalar register file] (round 6: measured peaks)
        assert 0 < v["frac_of_scalar_issue_peak"] < 1 and 0 < lo < hi and lo < 1, (k, v)
    assert r["kernels"]["parse"]["frac_of_scalar_issue_peak"] > r["kernels"]["recon"]["frac_of_scalar_issue_peak"] > r["kernels"]["deblock"]["frac_of_scalar_issue_peak"]


def test_grid_sharded_cannot_silently_measure_one_rank():
    """with N > 1 the `grid_sharded` section is only valid when the RCCL gather ran with N ranks and the peer-copy form reached
    other devices by peer access; anything else is named in `multi_gpu_check.failed` (and on stderr) instead of passing as an N-GPU time"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    good_photo = {"one_gpu_ms": 67.0, "rccl": {"ms": 20.0, "ranks": 8, "canvas_matches_one_gpu": True},
                  "peer_copy": {"ms": 21.0, "canvas_matches_one_gpu": True, "transport": {"shards_on_root_device": 1, "peer_access_shards": 7, "runtime_staged_shards": 0}}}
    ok, why = bench.check_grid_sharded({"wpp": good_photo, "pps_tiles_4x4": good_photo}, 8)
    assert ok and not why
    assert bench.check_grid_sharded({}, 1) == (True, [])                                   # one GPU: nothing to assert
    one_rank = dict(good_photo, rccl={"ms": 60.0, "ranks": 1, "canvas_matches_one_gpu": True})
    ok, why = bench.check_grid_sharded({"wpp": one_rank, "pps_tiles_4x4": good_photo}, 8)
    assert not ok and any("ranks=1" in w for w in why)
    staged = dict(good_photo, peer_copy=dict(good_photo["peer_copy"], transport={"shards_on_root_device": 1, "peer_access_shards": 0, "runtime_staged_shards": 7}))
    ok, why = bench.check_grid_sharded({"wpp": staged, "pps_tiles_4x4": good_photo}, 8)
    assert not ok and any("peer access" in w for w in why)
    ok, why = bench.check_grid_sharded({"error": "RuntimeError('x')"}, 2)
    assert not ok and len(why) >= 3
