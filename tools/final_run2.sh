#!/bin/bash
# round-end refresh after a host-side change: the default bench line again + rocprofv3 kernel statistics of the sequence tracks (one track, then 16 side by side)
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
timeout ${FINAL_BENCH_LIMIT:-900} python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
echo "bench rc=$?"; head -c 300 gpurun_out/final_bench.json; echo
( cd /tmp && export TMPDIR=/tmp
  SEQ_KIND=lowdelay timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_seq_prof -o p -- \
    python $GRAFT_REPO_ROOT/tools/sequence_fps.py 161 16 > $GRAFT_REPO_ROOT/gpurun_out/final_seq_prof.txt 2> $GRAFT_REPO_ROOT/gpurun_out/final_seq_prof.err
  echo "seq prof rc=$?"; grep fps $GRAFT_REPO_ROOT/gpurun_out/final_seq_prof.txt | cut -c1-250; f=$(find $GRAFT_REPO_ROOT/gpurun_out/final_seq_prof -name '*kernel_stats.csv' | head -1); head -14 $f | cut -c1-150 )
