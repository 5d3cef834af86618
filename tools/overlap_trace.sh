#!/bin/bash
# kernel timeline of the stage-overlap form (bench.py --parts 2): do the pixel kernels of batch k run beside k_parse of batch k+1?  (inside gpurun)
# usage: bash tools/overlap_trace.sh <tag> <pool waves> <stills>
tag=$1; pw=${2:-6144}; n=${3:-2048}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_trace
HIPDEC_POOL_WAVES=$pw timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/bench.py --only-main --steps 2 --warmup 1 --parts 2 --batch $n > $out.json 2> $out.err
f=$(find $out -name '*kernel_trace.csv' | head -1)
python - <<PY
import csv
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("hipdec::", "").replace("void ", "")[:28], r.get("Queue_Id", "")) for r in csv.DictReader(open("$f"))]
rows.sort()
t0 = rows[0][0]
big = [r for r in rows if r[1] - r[0] > 2_000_000]
for s, e, k, q in big[-24:]:
    print("%9.1f ms .. %9.1f ms  (%7.1f ms)  queue %s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, k))
PY
