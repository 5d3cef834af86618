#!/bin/bash
# GPU call: fps sweeps of pipelined chains under different numbers of HIP hardware queues.  usage (inside gpurun): [QUEUES="4 16"] [SWEEP=1:32,3:32] bash tools/seq_pipeline_sweep.sh <tag> [frames]   (SWEEP = chains in flight : look-ahead)
tag=$1; n=${2:-161}
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
for q in ${QUEUES:-default}; do
for kind in lowdelay unrestricted; do
  env $( [ $q = default ] || echo GPU_MAX_HW_QUEUES=$q ) SEQ_KIND=$kind SEQ_SWEEP=${SWEEP:-1:32,3:16,4:16,2:32,3:32,4:32} timeout 600 python tools/sequence_fps.py $n 16 2>&1 | grep -v "^\[libheif\|amdgpu.ids" | sed "s/^/hwq$q /; s/ x 1280x720 pictures (... KB per picture)//" | tee -a gpurun_out/${tag}_fps.txt
done; done
