#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the bench kernels at the benchmarked batch size -> gpurun_out/pmc_traffic.json
bash tools/prof_hbm_traffic.sh "$@" 2>&1 | tail -5
