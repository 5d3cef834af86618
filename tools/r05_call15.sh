#!/bin/bash
# round 5, GPU call 15: CABAC work pool or one wave per row for launch sets of a few thousand rows (16 tracks' chains)?  then the whole GPU tier and the bench line
mkdir -p gpurun_out
{
  echo "== look-ahead 32 (default), 65 pictures, 16 tracks: pool by default (>= 2048 rows) / HIPDEC_PARSE_POOL=0"
  SEQ_KIND=lowdelay timeout 300 python tools/sequence_fps.py 65 16 2>&1 | tail -1
  HIPDEC_PARSE_POOL=0 SEQ_KIND=lowdelay timeout 300 python tools/sequence_fps.py 65 16 2>&1 | tail -1
  echo "== 64 tracks: default / HIPDEC_PARSE_POOL=0"
  SEQ_KIND=lowdelay timeout 300 python tools/sequence_fps.py 65 64 2>&1 | tail -1
  HIPDEC_PARSE_POOL=0 SEQ_KIND=lowdelay timeout 300 python tools/sequence_fps.py 65 64 2>&1 | tail -1
  HIPDEC_CHAIN_TRACE=1 SEQ_KIND=lowdelay timeout 300 python tools/sequence_fps.py 65 64 2>&1 | grep "chain set" | tail -4
} 2>&1 | tee gpurun_out/c15_tracks.txt
python -c "import torch" 2>/dev/null
timeout 500 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/final_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/final_tests.log
timeout 700 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
echo "bench rc=$?"; tail -c 300 gpurun_out/final_bench.err; head -c 300 gpurun_out/final_bench.json; echo
