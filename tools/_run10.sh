cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in 6 8 12 24 34; do
HIPDEC_WAVES_PER_PICTURE=$w timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 256 > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("W=$w", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
done
