cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"; tail -2 gpurun_out/bench_final.err; cat gpurun_out/bench_final.json
bash tools/_prof.sh r1b --steps 2 --warmup 1 --no-cpu-baseline | tail -3
