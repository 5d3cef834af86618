cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "4 12 2048" "3 15 2048"; do
set -- $cfg
export HIPDEC_WAVES_PER_PICTURE=$1 HIPDEC_WPP_START_LAG=$2
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $3 --streams 2 > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "rc=$?"; tail -1 gpurun_out/bench_x.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("W $1 lag $2 batch $3:", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
done
