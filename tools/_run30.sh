cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_golden.py -q -m gpu -x 2>&1 | tail -3
for cfg in "0" "4" "8" "16"; do
if [ "$cfg" != "0" ]; then export HIPDEC_RECON_WAVES_PER_PICTURE=$cfg; else unset HIPDEC_RECON_WAVES_PER_PICTURE; fi
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 1024 --streams 1 > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; rc=$?; echo "rc=$rc"; grep -i "error" gpurun_out/bench_x.err | tail -1
[ $rc = 0 ] && python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json"))
print("reconW $cfg:", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
done
