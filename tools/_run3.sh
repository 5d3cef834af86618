cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for b in 256 512; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $b > gpurun_out/bench_b$b.json 2> gpurun_out/bench_b$b.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_b$b.json"))
print($b, d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
done
