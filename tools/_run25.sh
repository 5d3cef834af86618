cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
HIPDEC_PARSE_POOL=1 timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_golden.py -q -m gpu -x 2>&1 | tail -3
HIPDEC_PARSE_POOL=1 HIPDEC_POOL_YIELD=2 timeout 600 python -m pytest tests/test_decode_gpu.py -q -m gpu -x 2>&1 | tail -3
for cfg in "1024 2 4096" "1024 2 2048" "256 2 4096"; do
set -- $cfg
export HIPDEC_POOL_WAVES=$3
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $1 --streams $2 > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "rc=$?"; tail -1 gpurun_out/bench_x.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("batch $1 streams $2 pool waves $3:", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()}, d["single_still"]["ms"])
PY
done
