#!/bin/bash
# development GPU run (inside gpurun): usage: bash tools/gpu_stage.sh <tag> [tests] [bench] [benchmain] [pmc] [traffic] [prof]
tag=$1; shift
mkdir -p gpurun_out
for what in "$@"; do
  case $what in
    tests)     timeout ${TESTS_LIMIT:-420} python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${tag}_tests.log ;;
    bench)     timeout ${BENCH_LIMIT:-700} python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 400 gpurun_out/${tag}_bench.err; head -c 600 gpurun_out/${tag}_bench.json; echo ;;
    benchmain) timeout 300 python bench.py --only-main --steps 3 > gpurun_out/${tag}_benchmain.json 2> gpurun_out/${tag}_benchmain.err; echo "benchmain rc=$?"; tail -c 300 gpurun_out/${tag}_benchmain.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_benchmain.json")); print(d["value"], {k: v["avg_us"] for k, v in d["kernels"].items()})
except Exception as e: print("no line", e)
PY
 ;;
    single)    timeout 200 python bench.py --batch 1 --only-main --steps 3 > gpurun_out/${tag}_single.json 2> gpurun_out/${tag}_single.err; echo "single rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/${tag}_single.json')); print(d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})" ;;
    pmc)       bash tools/prof_parse_pmc.sh $tag --batch 512 > gpurun_out/${tag}_pmc.txt 2>&1; grep -E "k_parse|k_recon|pass" gpurun_out/${tag}_pmc.txt | cut -c1-420 ;;
    traffic)   bash tools/prof_hbm_traffic.sh 2>&1 | tail -3 ;;
    prof)      ( cd /tmp && export TMPDIR=/tmp
                 HIPDEC_SYNC_UPLOAD=1 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o p -- \
                   python $GRAFT_REPO_ROOT/bench.py --only-main --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.err
                 echo "prof rc=$?"; f=$(find $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 $f ) ;;
  esac
done
