cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "6 3072" "7 3072" "8 3072" "8 3584" "6 2560"; do
set -- $cfg
export HIPDEC_PARSE_OCCUPANCY=$1 HIPDEC_POOL_WAVES=$2
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 1024 --streams 2 > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; rc=$?; echo "rc=$rc"; grep -i "error" gpurun_out/bench_x.err | tail -1
[ $rc = 0 ] && python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("occ $1 pool waves $2:", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
done
