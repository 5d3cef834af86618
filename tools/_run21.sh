cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "1 2" "1 1" "0 2"; do
set -- $cfg
if [ "$1" = "1" ]; then export HIPDEC_DEBUG_PARSE_ONLY=1; else unset HIPDEC_DEBUG_PARSE_ONLY; fi
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 1024 --streams $2 > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("parse_only $1 streams $2:", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
done
