#!/bin/bash
# GPU call: pipelined look-ahead chains (HIPDEC_SEQ_PIPELINE) - the sequence tests with the pipeline on, then fps sweeps.  usage: bash tools/r06_pipeline_call.sh <tag> [frames]
tag=$1; n=${2:-161}
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
timeout 400 python -m pytest tests/test_sequence_pipeline_gpu.py -m gpu -q -x --timeout 300 > gpurun_out/${tag}_pipe_tests.log 2>&1; echo "pipeline tests rc=$?"; tail -3 gpurun_out/${tag}_pipe_tests.log
HIPDEC_SEQ_PIPELINE=3 timeout 500 python -m pytest tests/test_sequence_gpu.py tests/test_golden_sequences.py -m gpu -q --timeout 300 > gpurun_out/${tag}_seq_tests_p3.log 2>&1; echo "sequence tests (pipeline 3) rc=$?"; tail -3 gpurun_out/${tag}_seq_tests_p3.log
for kind in lowdelay unrestricted; do
  SEQ_KIND=$kind SEQ_SWEEP=${SWEEP_A:-1:32,2:32,3:32,3:16,4:16,1:64,2:64} timeout 600 python tools/sequence_fps.py $n 16 2>&1 | grep -v "^\[libheif" | tee -a gpurun_out/${tag}_fps.txt
  GPU_MAX_HW_QUEUES=8 SEQ_KIND=$kind SEQ_SWEEP=${SWEEP_B:-3:32,3:16,4:16,2:64} timeout 600 python tools/sequence_fps.py $n 16 2>&1 | grep -v "^\[libheif" | sed 's/^/hwq8 /' | tee -a gpurun_out/${tag}_fps.txt
done
