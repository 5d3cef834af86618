# PMC passes over the kernels of the bench workload: issue / stall / instruction-mix / instruction-cache counters, each group in its own
# rocprofv3 run (8 SQ slots per pass; no trace domains besides the kernel trace), every run under its own timeout.
# usage: bash tools/prof_parse_pmc.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
export HIPDEC_SYNC_UPLOAD=1     # keep the profiled process on one stream
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU" \
            "SQ_IFETCH SQ_WAIT_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
            "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  [ -n "$PMC_ONLY" ] && [ "$PMC_ONLY" != "$i" ] && { i=$((i+1)); continue; }
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i
  mkdir -p $out
  timeout 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/bench.py --only-main --steps 1 --warmup 0 "$@" > $out/bench.json 2> $out/bench.err
  echo "pass $i rc=$?"
  f=$(find $out -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - <<PY
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("$f")):
    agg[r["Kernel_Name"].split("(")[0][:36]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k in agg:
    if any(x in k for x in ("parse", "recon", "residual", "k_sao", "k_deblock")):
        print(k, " ".join("%s=%.5g"%(c,v) for c,v in sorted(agg[k].items())))
PY
  i=$((i+1))
done
# instructions per luma pixel of the issue-bound kernels -> gpurun_out/pmc_issue.json (copy to profiles/pmc_issue.json: bench.py's issue_roofline reads it)
python - "$tag" "$@" <<'PY'
import csv, collections, glob, json, os, subprocess, sys
root = os.environ["GRAFT_REPO_ROOT"]; tag = sys.argv[1]
f = glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_1" % tag, "**", "*counter_collection.csv"), recursive=True)
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
    try:
        bench = json.load(open(os.path.join(root, "gpurun_out", "pmc_%s_1" % tag, "bench.json")))
        n = bench["config"]["stills_per_step_per_gpu"]
    except Exception:
        n = 512
    px = n * 3840 * 2160
    out = {}
    for short in ("k_parse", "k_recon", "k_residual", "k_sao", "k_deblock"):
        tot = collections.defaultdict(float)
        for k, v in agg.items():
            if short in k:
                for c, x in v.items(): tot[c] += x
        if tot: out[short] = {"salu": round(tot["SQ_INSTS_SALU"] / px, 3), "valu": round(tot["SQ_INSTS_VALU"] / px, 3), "branch": round(tot["SQ_INSTS_BRANCH"] / px, 3), "lds": round(tot["SQ_INSTS_LDS"] / px, 3)}
    try: commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception: commit = os.environ.get("HIPDEC_COMMIT", "?")
    # effective clock (MI355X guide, DVFS): GRBM_GUI_ACTIVE of a dispatch / its duration, from the pass that collected it beside the kernel trace
    eff = {}
    try:
        f3 = glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_3" % tag, "**", "*counter_collection.csv"), recursive=True)
        kt = glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_3" % tag, "**", "*kernel_trace.csv"), recursive=True)
        if f3 and kt:
            dur = {}
            for r in csv.DictReader(open(kt[0])):
                dur[r.get("Dispatch_Id") or r.get("Correlation_Id")] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"])
            acc = collections.defaultdict(lambda: [0.0, 0.0])
            for r in csv.DictReader(open(f3[0])):
                if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
                d = dur.get(r.get("Dispatch_Id")) or dur.get(r.get("Correlation_Id"))
                if not d or d[0] <= 0: continue
                for short in ("k_parse", "k_recon", "k_residual", "k_sao", "k_deblock"):
                    if short in d[1]: acc[short][0] += float(r["Counter_Value"]); acc[short][1] += d[0]
            for short, (cyc, ns) in acc.items():
                ghz = cyc / ns
                for inst in (1, 8, 32, 64, 256):      # the counter is summed over its instances (XCDs / SEs): the divisor that lands on a plausible clock
                    if 0.8 <= ghz / inst <= 2.6: eff[short] = round(ghz / inst, 3); break
                else: eff[short + "_raw_cycles_per_ns"] = round(ghz, 3)
    except Exception as e:
        eff = {"error": repr(e)[:200]}
    doc = {"source": "rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH ... (tools/prof_parse_pmc.sh, own pass, kernel trace only) over `bench.py --only-main --steps 1 --warmup 0 " + " ".join(sys.argv[2:]) + "` on MI355X",
           "commit": commit, "stills_per_step": n, "insts_per_px": out, "effective_clock_ghz": eff}
    json.dump(doc, open(os.path.join(root, "gpurun_out", "pmc_issue.json"), "w"), indent=1)
    print(json.dumps(out))
PY
