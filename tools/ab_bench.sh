#!/bin/bash
# inside gpurun: the main workload's kernel times with the in-tree library and with another branch's build (tools/ab_build.sh), then - with
# TESTS=1 - the GPU tier under that build; SINGLE=1 adds one 4K still, FROM_HOST=1 the from-host-bytes `value`.
# usage: [TESTS=1] [SINGLE=1] [FROM_HOST=1] bash tools/ab_bench.sh [branch] [bench args]   (default branch: candidates)
br=${1:-candidates}; shift || true
alt=build/ab/$br/libheif_amd/libheifhip.so
mkdir -p gpurun_out
export HIPDEC_DEV_AB=1   # libheif_amd/_capi.py honours HIPDEC_LIBRARY only with this
[ -f $alt ] || { echo "no $alt: run tools/ab_build.sh $br first"; exit 1; }
for which in tree $br; do
  lib=""; [ $which = tree ] || lib=$PWD/$alt
  HIPDEC_LIBRARY=$lib timeout 200 python bench.py --only-main --steps 3 --warmup 1 "$@" > gpurun_out/ab_$which.json 2> gpurun_out/ab_$which.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ab_$which.json")); print("%-12s" % "$which", d["value"], {k: v["avg_us"] for k, v in d["kernels"].items()})
except Exception as e: print("$which: no line", e)
PY
done
if [ -n "$SINGLE" ]; then   # one 4K still (BASELINE config 2): the latency-bound form
  for which in tree $br; do
    lib=""; [ $which = tree ] || lib=$PWD/$alt
    HIPDEC_LIBRARY=$lib timeout 120 python bench.py --batch 1 --only-main --steps 3 > gpurun_out/ab_single_$which.json 2> gpurun_out/ab_single_$which.err
    python -c "
import json; d=json.load(open('gpurun_out/ab_single_$which.json')); print('%-12s' % '$which', 'one still', d['ms_per_step'], 'ms', {k: v['avg_us'] for k, v in d['kernels'].items()})"
  done
fi
if [ -n "$FROM_HOST" ]; then   # `value` (from host bytes) needs the default line's first section: both libraries, without the extras
  for which in tree $br; do
    lib=""; [ $which = tree ] || lib=$PWD/$alt
    HIPDEC_LIBRARY=$lib timeout 300 python bench.py --no-extras --no-dropin --no-cpu-baseline --no-grid-sharded --steps 3 > gpurun_out/ab_host_$which.json 2> gpurun_out/ab_host_$which.err
    python -c "
import json; d=json.load(open('gpurun_out/ab_host_$which.json')); print('%-12s' % '$which', 'value', d['value'], 'resident', d.get('value_resident'), d['ms_per_step'])"
  done
fi
if [ -n "$TESTS" ]; then
  HIPDEC_LIBRARY=$PWD/$alt timeout ${TESTS_LIMIT:-400} python -m pytest tests -m gpu -q -x > gpurun_out/ab_tests_$br.log 2>&1; echo "tests under $br rc=$?"; tail -3 gpurun_out/ab_tests_$br.log
fi
