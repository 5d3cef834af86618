#!/bin/bash
# round 5, GPU call 13: the chains of tracks decoded side by side as ONE launch set (decoder.hip: ChainCoalescer): sequence tier + fps of 16 / 64 tracks
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_sequence_gpu.py -m gpu -q --timeout 200 -x > gpurun_out/c13_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c13_tests.log | cut -c1-300
{
  for kind in lowdelay unrestricted; do
    SEQ_KIND=$kind timeout 200 python tools/sequence_fps.py 33 16 2>&1 | tail -1
  done
  echo "== 64 tracks"
  SEQ_KIND=lowdelay timeout 200 python tools/sequence_fps.py 33 64 2>&1 | tail -1
  echo "== trace of the 16-track run"
  HIPDEC_CHAIN_TRACE=1 SEQ_KIND=lowdelay timeout 200 python tools/sequence_fps.py 33 16 2>&1 | grep -E "chain set|tracks side" | tail -12
  echo "== every chain on its own (HIPDEC_CHAIN_WINDOW_US=0)"
  HIPDEC_CHAIN_WINDOW_US=0 SEQ_KIND=lowdelay timeout 200 python tools/sequence_fps.py 33 16 2>&1 | tail -1
} 2>&1 | tee gpurun_out/c13_tracks.txt
