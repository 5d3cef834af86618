#!/bin/bash
# round-end GPU run fitted into a given number of seconds (inside gpurun): tests, rocprofv3 kernel stats, PMC issue counters, HBM traffic, main workload,
# and - if the time left allows - the default bench line.  usage: final_fit.sh <tag> <seconds>
tag=$1; total=${2:-320}; t0=$(date +%s)
TESTS_LIMIT=120 bash tools/gpu_stage.sh $tag tests
bash tools/gpu_stage.sh $tag prof pmc traffic benchmain
left=$(( total - ($(date +%s) - t0) - 8 ))
echo "time left for the default bench line: $left s"
if [ $left -ge 130 ]; then BENCH_LIMIT=$left bash tools/gpu_stage.sh $tag bench; else echo "bench skipped"; fi
