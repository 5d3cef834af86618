cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
HIPDEC_PARSE_OCCUPANCY=8 timeout 600 python -m pytest tests/test_decode_gpu.py -q -m gpu -x 2>&1 | tail -2
for cfg in "7 0" "8 0" "8 12"; do
set -- $cfg
export HIPDEC_PARSE_OCCUPANCY=$1
if [ "$2" != "0" ]; then export HIPDEC_WAVES_PER_PICTURE=$2; else unset HIPDEC_WAVES_PER_PICTURE; fi
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 1024 --streams 2 > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("occ $1 W $2:", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()}, d["single_still"]["ms"])
PY
done
