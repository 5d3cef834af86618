#!/bin/bash
# What do the streaming kernels wait for?    PMC passes over the bench kernels, one counter group per rocprofv3 run (kernel
# trace only beside the counters), per-kernel totals -> gpurun_out/<tag>_wait_breakdown.txt.  A group with a counter this rocprofv3 does not know
# fails on its own and is reported; the others still run.   usage: bash tools/prof_wait_breakdown.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
export HIPDEC_SYNC_UPLOAD=1
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_wait_breakdown.txt
: > $out
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/wb_${tag}_$i
  rm -rf $d; mkdir -p $d
  timeout 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --only-main --steps 1 --warmup 0 "$@" > $d/bench.json 2> $d/bench.err
  rc=$?
  echo "== group $i rc=$rc: $ctrs" >> $out
  f=$(find $d -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python - "$f" >> $out <<'PY'
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(agg):
    if any(x in k for x in ("parse", "recon", "residual", "k_sao", "k_deblock")):
        print("  %-42s" % k, " ".join("%s=%.5g" % (c, v) for c, v in sorted(agg[k].items())))
PY
  else tail -3 $d/bench.err | cut -c1-300 >> $out; fi
  i=$((i+1))
done
cat $out
