#!/bin/bash
# inside gpurun: several builds of the library (tools/ab_build.sh <branch> each) against the in-tree one in ONE call: the main workload's kernel
# times, one 4K still, optionally `value` from host bytes and the GPU tier under the last build named.
# usage: [TESTS=1] [FROM_HOST=1] [STEPS=4] bash tools/ab_multi.sh <branch> [<branch> ...]
mkdir -p gpurun_out
export HIPDEC_DEV_AB=1   # libheif_amd/_capi.py honours HIPDEC_LIBRARY only with this
libs="tree $*"; last=""
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print("%-12s" % sys.argv[1], d.get("value_resident", d["value"]), "ms/step", d["ms_per_step"], {k: round(v["avg_us"] / 1e3, 2) for k, v in d["kernels"].items()})
except Exception as e: print(sys.argv[1], "no line", e)
PY
}
for rep in 1 2; do
for which in $libs; do
  lib=""; [ $which = tree ] || lib=$PWD/build/ab/$which/libheif_amd/libheifhip.so
  HIPDEC_LIBRARY=$lib timeout 200 python bench.py --only-main --steps ${STEPS:-4} --warmup 1 > gpurun_out/abm_${which}_$rep.json 2> gpurun_out/abm_${which}_$rep.err
  show "$which/$rep" gpurun_out/abm_${which}_$rep.json
done; done
for which in $libs; do
  lib=""; [ $which = tree ] || lib=$PWD/build/ab/$which/libheif_amd/libheifhip.so
  HIPDEC_LIBRARY=$lib timeout 120 python bench.py --batch 1 --only-main --steps 5 > gpurun_out/abm_single_$which.json 2> gpurun_out/abm_single_$which.err
  show "$which/still" gpurun_out/abm_single_$which.json
  last=$which
done
if [ -n "$FROM_HOST" ]; then
  for which in tree $last; do
    lib=""; [ $which = tree ] || lib=$PWD/build/ab/$which/libheif_amd/libheifhip.so
    HIPDEC_LIBRARY=$lib timeout 300 python bench.py --no-extras --no-dropin --no-cpu-baseline --no-grid-sharded --steps 3 > gpurun_out/abm_host_$which.json 2> gpurun_out/abm_host_$which.err
    python -c "
import json; d=json.load(open('gpurun_out/abm_host_$which.json')); print('%-12s' % '$which', 'value', d['value'], 'resident', d.get('value_resident'), d['ms_per_step'], 'verified', d.get('verified'))"
  done
fi
if [ -n "$TESTS" ]; then
  HIPDEC_LIBRARY=$PWD/build/ab/$last/libheif_amd/libheifhip.so timeout ${TESTS_LIMIT:-400} python -m pytest tests -m gpu -q > gpurun_out/abm_tests_$last.log 2>&1; echo "tests under $last rc=$?"; tail -5 gpurun_out/abm_tests_$last.log
fi
