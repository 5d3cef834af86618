cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "1024 2 4096" "1024 2 3072" "1024 2 1536" "1024 1 4096" "512 2 2048"; do
set -- $cfg
export HIPDEC_POOL_WAVES=$3
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $1 --streams $2 > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "rc=$?"; grep -i "error" gpurun_out/bench_x.err | tail -1
python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("batch $1 streams $2 pool waves $3:", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
done
