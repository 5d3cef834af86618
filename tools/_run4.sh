cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 800 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/t6.log; cat gpurun_out/t6.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --workload grid8k --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_grid.json 2> gpurun_out/bench_grid.err; echo "grid rc=$?"; tail -2 gpurun_out/bench_grid.err; cat gpurun_out/bench_grid.json
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default rc=$?"; tail -2 gpurun_out/bench_default.err; cat gpurun_out/bench_default.json
