# HBM traffic of the bench kernels at the benchmarked batch size: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X guide: TCC
# slots), each under its own timeout; writes gpurun_out/pmc_traffic.json (copy to profiles/pmc_traffic.json).  usage: bash tools/prof_hbm_traffic.sh [bench args]
cd /tmp && export TMPDIR=/tmp
export HIPDEC_SYNC_UPLOAD=1      # keep the profiled process on one stream (rocprofv3 counter collection + cross-stream event waits hung)
for c in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/bench.py --only-main --steps 1 --warmup 0 "$@" > $out/bench.json 2> $out/bench.err
  echo "$c rc=$?"
done
python - "$@" <<'PY'
import csv, collections, glob, json, os, sys
root = os.environ["GRAFT_REPO_ROOT"]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(root, "gpurun_out", "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"]] += float(r["Counter_Value"])
    res[c] = agg
bench = json.load(open(os.path.join(root, "gpurun_out", "pmc_FETCH_SIZE", "bench.json")))
px = bench["config"]["stills_per_step_per_gpu"] * 3840 * 2160
names = {"k_parse": "k_parse", "k_residual": "k_residual", "k_recon": "k_recon", "k_deblock": "k_deblock", "k_sao": "k_sao", "k_ycbcr_to_rgb": "k_ycbcr_to_rgb"}
def per_px(agg):
    out = {}
    for short in names:
        tot = sum(v for k, v in agg.items() if short in k)
        out[short] = round(tot * 1024.0 / px, 4)      # counter unit: KiB
    return out
# gfx950 correction (MI355X guide, HBM section; calibrated for this project's access widths by tools/ubench/fetch_calib.hip, profiles/r05_fetch_calibration.txt):
# FETCH_SIZE reports exactly HALF of the bytes read - for 1-, 4- and 16-byte-per-lane loads alike - and WRITE_SIZE the bytes written
f_raw, w = per_px(res["FETCH_SIZE"]), per_px(res["WRITE_SIZE"])
f = {k: round(2.0 * v, 4) for k, v in f_raw.items()}
import subprocess
try: commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception: commit = os.environ.get("HIPDEC_COMMIT", "?")
doc = {"commit": commit, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_hbm_traffic.sh) over `bench.py --only-main --steps 1 --warmup 0 " + " ".join(sys.argv[1:]) + "` on MI355X; counter unit KiB; FETCH_SIZE DOUBLED (calibrated: profiles/r05_fetch_calibration.txt), WRITE_SIZE as reported",
       "workload": "still4k", "qp": 27, "stills_per_step": bench["config"]["stills_per_step_per_gpu"],
       "fetch_bytes_per_px": f, "fetch_size_counter_bytes_per_px": f_raw, "write_bytes_per_px": w, "bytes_per_px": {k: round(f[k] + w[k], 4) for k in f}}
json.dump(doc, open(os.path.join(root, "gpurun_out", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(doc["bytes_per_px"]))
PY
