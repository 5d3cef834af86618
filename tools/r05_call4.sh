#!/bin/bash
# round 5, GPU call 4: GPU tier; sequence fps + kernel stats with the staged k_motion; drop-in throughput (planes, RGB through the patched libheif)
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/c4_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/c4_tests.log | cut -c1-300
for k in 8 16 32; do
  echo "== HIPDEC_SEQ_LOOKAHEAD=$k"; HIPDEC_SEQ_LOOKAHEAD=$k timeout 300 python tools/sequence_fps.py 33 16 2>&1 | tail -3
done > gpurun_out/c4_seqfps.txt 2>&1
cat gpurun_out/c4_seqfps.txt
( cd /tmp && export TMPDIR=/tmp
  SEQ_KIND=lowdelay timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c4_seqprof -o p -- \
     python $GRAFT_REPO_ROOT/tools/sequence_fps.py 33 1 > $GRAFT_REPO_ROOT/gpurun_out/c4_seqprof.txt 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/c4_seqprof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 $f | cut -c1-160 )
for mode in "planes libheif.so" "rgb libheif_hipcolor.so" "planes libheif_hipcolor.so"; do
  set -- $mode
  flag=""; [ $1 = rgb ] && flag="--rgb"
  echo "== dropin $1 through $2"
  HIPDEC_IMAGE_OPS_TIMING=1 timeout 200 python tools/dropin_throughput.py --threads 256,1024 --seconds 5 --libheif $2 $flag --json 2> gpurun_out/c4_dropin_$1_$2.err | python -c "
import json,sys
d=json.load(sys.stdin); print([(r.get('threads'), r.get('mpixel_s'), r.get('stills_per_launch_set'), r.get('error')) for r in d['runs']])"
  grep "color_convert" gpurun_out/c4_dropin_$1_$2.err | tail -2 | cut -c1-300
done 2>&1 | tee gpurun_out/c4_dropin.txt
timeout 120 python tools/plugin_grid_timing.py 2>&1 | tail -6 | tee gpurun_out/c4_grid_timing.txt
