cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
nproc
for args in "--batch 1024 --streams 2 --distinct 16" "--batch 1024 --streams 2 --distinct 64"; do
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; rc=$?; echo "rc=$rc"; grep -i "error" gpurun_out/bench_x.err | tail -1
[ $rc = 0 ] && python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("$args:", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()}, d["config"]["bitstream_bytes_per_px"])
PY
done
