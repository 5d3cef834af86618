#!/bin/bash
# round 5, GPU call 18: residuals of inter coded units added by k_mc (fully parallel), k_recon's wavefront only reconstructs the intra blocks of P / B pictures
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_sequence_gpu.py -m gpu -q --timeout 150 -x > gpurun_out/c18_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/c18_tests.log | cut -c1-300
timeout 200 python -c "
import json, bench
print(json.dumps(bench.sequence_tracks()))" > gpurun_out/c18_sequence_tracks.json 2> gpurun_out/c18_sequence_tracks.err; echo "sequence_tracks rc=$?"; cut -c1-1500 gpurun_out/c18_sequence_tracks.json | tr ',' '\n' | grep -E "fps|launch_sets"
echo "== the form before (HIPDEC_INTER_RECON_PER_BLOCK=1), same box"
HIPDEC_INTER_RECON_PER_BLOCK=1 SEQ_KIND=lowdelay timeout 200 python tools/sequence_fps.py 65 16 2>&1 | tail -1 | tee gpurun_out/c18_ab.txt
SEQ_KIND=lowdelay timeout 200 python tools/sequence_fps.py 65 16 2>&1 | tail -1 | tee -a gpurun_out/c18_ab.txt
cd /tmp && export TMPDIR=/tmp
SEQ_KIND=lowdelay timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c18_seqprof -o p -- \
  python $GRAFT_REPO_ROOT/tools/sequence_fps.py 65 16 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/c18_seqprof -name '*kernel_stats.csv' | head -1); head -8 "$f" | cut -c1-160
find $GRAFT_REPO_ROOT/gpurun_out/c18_seqprof -name '*kernel_trace.csv' -delete
