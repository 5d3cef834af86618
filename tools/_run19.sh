cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_golden.py -q -m gpu -x 2>&1 | tail -2
./build/cabac_ubench 200000 2>&1 | grep -E "decision bin \(ctx cyc|bypass"
for args in "--batch 1024 --streams 2" "--batch 16 --streams 1"; do
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $args > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("$args", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()}, d["single_still"]["ms"])
PY
done
