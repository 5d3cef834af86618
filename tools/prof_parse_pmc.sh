# PMC passes over the kernels of the bench workload: issue / stall / instruction-mix / instruction-cache counters, each group in its own
# rocprofv3 run (8 SQ slots per pass; no trace domains besides the kernel trace), every run under its own timeout.
# usage: bash tools/prof_parse_pmc.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
export HIPDEC_SYNC_UPLOAD=1     # keep the profiled process on one stream
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU" \
            "SQ_IFETCH SQ_WAIT_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
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
    if "parse" in k or "recon" in k or "residual" in k:
        print(k, " ".join("%s=%.5g"%(c,v) for c,v in sorted(agg[k].items())))
PY
  i=$((i+1))
done
