cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for args in "--workload still1080 --batch 1024 --enc wpp=0" "--workload still1080 --batch 4096 --enc wpp=0" "--workload still1080 --batch 4096"; do
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $args > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("$args", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()}, d["config"]["substreams_per_still"])
PY
done
