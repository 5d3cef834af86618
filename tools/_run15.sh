cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_decode_gpu.py -q -m gpu -x 2>&1 | tail -2
for cfg in "1024 2 0" "1024 2 6" "256 1 0" "256 1 2" "512 2 0"; do
set -- $cfg
if [ "$3" != "0" ]; then export HIPDEC_WPP_START_LAG=$3; else unset HIPDEC_WPP_START_LAG; fi
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $1 --streams $2 > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("batch $1 streams $2 lag $3:", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
done
