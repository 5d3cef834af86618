cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"; grep -i error gpurun_out/bench_final.err | tail -2; cat gpurun_out/bench_final.json
bash tools/_prof.sh r1f --steps 2 --warmup 1 --no-cpu-baseline | tail -12 | cut -c1-170
