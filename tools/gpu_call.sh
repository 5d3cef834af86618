#!/bin/bash
# One gpurun call = a list of stages (replaces the per-call one-off scripts of round 5).  usage (inside gpurun): bash tools/gpu_call.sh <tag> <stage> ...
# Stages: every stage of tools/gpu_stage.sh (tests bench benchmain single pmc traffic prof), plus
#   ubench_issue          tools/ubench/issue_model.hip (prebuilt: build/ubench/issue_model) -> gpurun_out/<tag>_issue_model.txt
#   att_try               does `rocprofv3 --att` work on this box at all?  (the trace-decoder library is not in the image)
#   ab:<name>[@ENV=V..][,<name>..]  main workload under build/ab/<name>/libheif_amd/libheifhip.so variants (tools/ab_variant.sh), tree first;
#                         "tree@ENV=V" runs the tree's library with that environment
#   absingle:<names>      one 4K still under the same variants
#   abtests:<name>        the GPU tier under that variant
#   wait                  tools/prof_wait_breakdown.sh
tag=$1; shift
mkdir -p gpurun_out
export HIPDEC_DEV_AB=1
python -c "import torch" 2>/dev/null
for what in "$@"; do
  case $what in
    ubench_issue) timeout 300 build/ubench/issue_model ${UBENCH_ITERS:-4000} ${UBENCH_FIRST:-0} > gpurun_out/${tag}_issue_model.txt 2>&1; echo "ubench_issue rc=$?"; cat gpurun_out/${tag}_issue_model.txt ;;
    att_try) ( cd /tmp && export TMPDIR=/tmp
               timeout 120 rocprofv3 --att --att-target-cu 1 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_att -- $GRAFT_REPO_ROOT/build/ubench/issue_model 200 12 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_att.log 2>&1
               echo "att rc=$?"; tail -15 $GRAFT_REPO_ROOT/gpurun_out/${tag}_att.log; find $GRAFT_REPO_ROOT/gpurun_out/${tag}_att -type f | head -20; find / \( -name "*trace-decoder*" -o -name "*trace_decoder*.so*" \) 2>/dev/null | head ) ;;
    ab:*) for spec in tree $(echo ${what#ab:} | tr ',' ' '); do     # <name>[@ENV=VALUE[@ENV=VALUE..]]: environment of that variant's run
            name=${spec%%@*}; envs=$(echo "${spec#$name}" | tr '@' ' ')
            lib=""; [ $name = tree ] || lib=$PWD/build/ab/$name/libheif_amd/libheifhip.so
            name=$(echo $spec | tr '@=' '__')
            env $envs HIPDEC_LIBRARY=$lib timeout 240 python bench.py --only-main --steps ${AB_STEPS:-3} --warmup 1 ${AB_ARGS} > gpurun_out/${tag}_ab_$name.json 2> gpurun_out/${tag}_ab_$name.err
            python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_ab_$name.json")); print("%-14s" % "$name", d["value"], {k: round(v["avg_us"]) for k, v in d["kernels"].items()}, "verified", d.get("verified"))
except Exception as e: print("$name: no line", e)
PY
          done ;;
    absingle:*) for name in tree $(echo ${what#absingle:} | tr ',' ' '); do
            lib=""; [ $name = tree ] || lib=$PWD/build/ab/$name/libheif_amd/libheifhip.so
            HIPDEC_LIBRARY=$lib timeout 120 python bench.py --batch 1 --only-main --steps 3 > gpurun_out/${tag}_absingle_$name.json 2> gpurun_out/${tag}_absingle_$name.err
            python -c "
import json; d=json.load(open('gpurun_out/${tag}_absingle_$name.json')); print('%-14s' % '$name', 'one still', d['ms_per_step'], 'ms', {k: round(v['avg_us']) for k, v in d['kernels'].items()})" ;
          done ;;
    abtests:*) name=${what#abtests:}
            HIPDEC_LIBRARY=$PWD/build/ab/$name/libheif_amd/libheifhip.so timeout ${TESTS_LIMIT:-420} python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_abtests_$name.log 2>&1; echo "tests under $name rc=$?"; tail -3 gpurun_out/${tag}_abtests_$name.log ;;
    overlap) # pixel stages of batch k beside the CABAC parse of batch k+1, FULL-size batches, pool of 8 / 7 / 6 / 5 waves per SIMD
          for spec in "1:1536:8192" "1:3072:8192" "2:3072:8192" "2:3072:7168" "2:3072:6144" "2:3072:5120"; do
            parts=${spec%%:*}; rest=${spec#*:}; n=${rest%%:*}; pw=${rest#*:}
            HIPDEC_POOL_WAVES=$pw timeout 400 python bench.py --only-main --steps 3 --warmup 1 --parts $parts --batch $n > gpurun_out/${tag}_overlap_${parts}_${n}_${pw}.json 2> gpurun_out/${tag}_overlap_${parts}_${n}_${pw}.err
            python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_overlap_${parts}_${n}_${pw}.json")); print("parts $parts  stills $n  pool waves $pw :", d["value"], "Mpx/s", d["ms_per_step"], "ms/step", {k: round(v["avg_us"]) for k, v in d["kernels"].items()})
except Exception as e: print("$spec: no line", e)
PY
          done ;;
    dropin) python tools/dropin_throughput.py --threads ${DROPIN_THREADS:-1,8,32,64,256} --seconds 3 > gpurun_out/${tag}_dropin_planes.txt 2> gpurun_out/${tag}_dropin_planes.err; tail -8 gpurun_out/${tag}_dropin_planes.txt
            python tools/dropin_throughput.py --threads ${DROPIN_RGB_THREADS:-64,256} --seconds 3 --rgb --libheif libheif_hipcolor.so > gpurun_out/${tag}_dropin_rgb.txt 2> gpurun_out/${tag}_dropin_rgb.err; tail -4 gpurun_out/${tag}_dropin_rgb.txt; grep -h "on average" gpurun_out/${tag}_dropin_*.err | tail -8 ;;
    wait) bash tools/prof_wait_breakdown.sh $tag --batch 512 2>&1 | tail -40 ;;
    *) bash tools/gpu_stage.sh $tag $what ;;
  esac
done
