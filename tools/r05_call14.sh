#!/bin/bash
# round 5, GPU call 14: all motion steps of a chain as ONE k_motion launch (pictures follow their collocated picture at a 2-CTB distance) against one launch per step
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_sequence_gpu.py -m gpu -q --timeout 200 -x > gpurun_out/c14_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c14_tests.log | cut -c1-300
{
  echo "== one launch (default)"
  for kind in lowdelay unrestricted; do SEQ_KIND=$kind timeout 200 python tools/sequence_fps.py 33 16 2>&1 | tail -1; done
  echo "== one launch per motion step (HIPDEC_CHAIN_MOTION_STEPS=1)"
  for kind in lowdelay unrestricted; do HIPDEC_CHAIN_MOTION_STEPS=1 SEQ_KIND=$kind timeout 200 python tools/sequence_fps.py 33 16 2>&1 | tail -1; done
  echo "== look-ahead 32, 65 pictures"
  for kind in lowdelay unrestricted; do HIPDEC_SEQ_LOOKAHEAD=32 SEQ_KIND=$kind timeout 300 python tools/sequence_fps.py 65 16 2>&1 | tail -1; done
  echo "== trace"
  HIPDEC_CHAIN_TRACE=1 SEQ_KIND=lowdelay timeout 200 python tools/sequence_fps.py 33 16 2>&1 | grep -E "chain set" | tail -6
} 2>&1 | tee gpurun_out/c14_tracks.txt
# repeat the threaded test a few times: the cross-picture hand-over is a memory-ordering protocol
for i in 1 2 3; do timeout 200 python -m pytest tests/test_sequence_gpu.py -m gpu -q --timeout 150 -k "side_by_side or b_tmvp or through_libheif" 2>&1 | tail -1; done | tee -a gpurun_out/c14_tracks.txt
