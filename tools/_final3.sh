cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2; do
timeout 900 python bench.py > gpurun_out/bench_final$i.json 2> gpurun_out/bench_final$i.err; echo "bench rc=$?"; grep -i error gpurun_out/bench_final$i.err | tail -2; python -c "
import json; d=json.load(open('gpurun_out/bench_final$i.json')); print(d['value'], d['ms_per_step'], d['cpu_baseline']['value'])"
done
bash tools/_prof.sh r1h --steps 3 --warmup 1 --no-cpu-baseline | tail -12 | cut -c1-170
