#!/bin/bash
# kernel timeline of one 720p track (inside gpurun): usage: bash tools/seq_timeline.sh <tag> <pipeline> <lookahead> [frames]
tag=$1; export HIPDEC_SEQ_PIPELINE=$2 HIPDEC_SEQ_LOOKAHEAD=$3; n=${4:-97}
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_trace
HIPDEC_CHAIN_TRACE=1 timeout 300 python tools/seq_single_track.py $n 2>&1 | grep -v "^\[libheif" | tail -40
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/tools/seq_single_track.py $n > $out.log 2>&1
f=$(find $out -name '*kernel_trace.csv' | head -1)
python - <<PY
import csv
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("hipdec::", "").replace("void ", "")[:24], r.get("Queue_Id", ""), r.get("Stream_Id", "")) for r in csv.DictReader(open("$f"))]
rows.sort()
big = [r for r in rows if "parse" in r[2] or "motion" in r[2]]
# the second play: the last parse_inter launches
sel = big[-14:]
t0 = sel[0][0]
for s, e, k, q, st in sel:
    print("%9.1f ms .. %9.1f ms  (%7.1f ms)  queue %s stream %s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, st, k))
rec = [r for r in rows if "recon8_inter" in r[2] and r[0] >= t0]
print("k_recon8_inter launches after that point:", len(rec), " first %.1f ms last %.1f ms" % ((rec[0][0] - t0) / 1e6, (rec[-1][1] - t0) / 1e6))
# gaps between consecutive recon launches > 5 ms
prev = None
for s, e, k, q, st in rec:
    if prev is not None and s - prev > 5_000_000: print("   gap of %.1f ms before the launch at %.1f ms" % ((s - prev) / 1e6, (s - t0) / 1e6))
    prev = e
PY
