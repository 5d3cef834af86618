#!/bin/bash
# round 5, GPU call 19: the default bench line at the round's final code
mkdir -p gpurun_out
timeout 270 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
echo "bench rc=$?"; head -c 200 gpurun_out/final_bench.json; echo
