cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_decode_gpu.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/t3.log; cat gpurun_out/t3.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "rc=$?"
tail -3 gpurun_out/bench2.err; cat gpurun_out/bench2.json
