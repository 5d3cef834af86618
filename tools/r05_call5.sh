#!/bin/bash
# round 5, GPU call 5: GPU tier, wait breakdown of the streaming kernels, effective clock + instruction counts, FETCH_SIZE / WRITE_SIZE calibration
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/c5_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/c5_tests.log | cut -c1-300
bash tools/prof_wait_breakdown.sh c5 --batch 512 > /dev/null 2>&1; cat gpurun_out/c5_wait_breakdown.txt | cut -c1-600
PMC_ONLY=1 bash tools/prof_parse_pmc.sh c5 --batch 512 > gpurun_out/c5_pmc1.txt 2>&1
PMC_ONLY=3 bash tools/prof_parse_pmc.sh c5 --batch 512 > gpurun_out/c5_pmc3.txt 2>&1; tail -3 gpurun_out/c5_pmc3.txt | cut -c1-600; cat gpurun_out/pmc_issue.json | cut -c1-1200
( cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$GRAFT_REPO_ROOT/gpurun_out/c5_calib_$c; rm -rf $d; mkdir -p $d
    timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- $GRAFT_REPO_ROOT/build/fetch_calib > $d/out.txt 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - "$f" $c <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and "calib" in r["Kernel_Name"]:
        v = float(r["Counter_Value"])
        print("%-11s %-60s %.6g KiB -> %.4f of the 2^30 bytes moved" % (sys.argv[2], r["Kernel_Name"][:60], v, v * 1024 / 2**30))
PY
  done ) 2>&1 | tee gpurun_out/c5_fetch_calibration.txt
