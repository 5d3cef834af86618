cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for args in "--batch 1024 --streams 3" "--batch 1024 --streams 4" "--batch 1536 --streams 3"; do
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $args > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("$args:", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
done
