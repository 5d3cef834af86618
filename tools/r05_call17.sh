#!/bin/bash
# round 5, GPU call 17: rocprofv3 kernel statistics of sequence tracks at the final code (one IPPP track and 16 side by side, look-ahead 32)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
SEQ_KIND=lowdelay timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c17_seqprof -o p -- \
  python $GRAFT_REPO_ROOT/tools/sequence_fps.py 65 16 > $GRAFT_REPO_ROOT/gpurun_out/c17_seq.txt 2>&1
echo "rc=$?"; tail -1 $GRAFT_REPO_ROOT/gpurun_out/c17_seq.txt
f=$(find $GRAFT_REPO_ROOT/gpurun_out/c17_seqprof -name '*kernel_stats.csv' | head -1); echo $f; head -12 "$f" | cut -c1-200
find $GRAFT_REPO_ROOT/gpurun_out/c17_seqprof -name '*kernel_trace.csv' -size +30M -delete
