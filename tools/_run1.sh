cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "rc=$?" >> gpurun_out/bench1.err
tail -3 gpurun_out/bench1.err; cat gpurun_out/bench1.json
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/t2.log; cat gpurun_out/t2.log
