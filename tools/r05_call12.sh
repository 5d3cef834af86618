#!/bin/bash
# round 5, GPU call 12: do 16 tracks side by side scale with more HIP hardware queues?  + the sequence tier after the shared-set reference copy
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_sequence_gpu.py -m gpu -q --timeout 200 > gpurun_out/c12_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/c12_tests.log | cut -c1-200
for q in default 8 16 32; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  if [ $q = default ]; then SEQ_KIND=lowdelay timeout 200 python tools/sequence_fps.py 33 16 2>&1 | tail -1
  else GPU_MAX_HW_QUEUES=$q SEQ_KIND=lowdelay timeout 200 python tools/sequence_fps.py 33 16 2>&1 | tail -1; fi
done 2>&1 | tee gpurun_out/c12_hwqueues.txt
GPU_MAX_HW_QUEUES=16 SEQ_KIND=lowdelay timeout 200 python tools/sequence_fps.py 33 64 2>&1 | tail -1 | tee -a gpurun_out/c12_hwqueues.txt
