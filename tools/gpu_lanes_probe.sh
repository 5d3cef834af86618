#!/bin/bash
# GPU probe of the lane-per-substream parser: parity tests with it forced on, then the headline bench with either parser.
mkdir -p gpurun_out
export HIPDEC_PARSE_LANES=1
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_full_shape_gpu.py -m gpu -q -x > gpurun_out/lanes_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/lanes_tests.log
tail -5 gpurun_out/lanes_tests.log
for n in 2048 1024 256; do
  HIPDEC_PARSE_LANES=1 timeout 600 python bench.py --only-main --no-extras --no-cpu-baseline --steps 2 --warmup 1 --batch $n > gpurun_out/lanes_bench_$n.json 2> gpurun_out/lanes_bench_$n.err
  echo "lanes n=$n rc=$?"; tail -c 1500 gpurun_out/lanes_bench_$n.json
done
HIPDEC_PARSE_LANES=0 timeout 600 python bench.py --only-main --no-extras --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/waves_bench_2048.json 2> gpurun_out/waves_bench_2048.err
echo "waves rc=$?"; tail -c 1500 gpurun_out/waves_bench_2048.json
