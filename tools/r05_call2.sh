#!/bin/bash
# round 5, GPU call 2: the GPU tier with the look-ahead, sequence fps per look-ahead, k_recon (h) alone, the parser at 6 waves per SIMD
mkdir -p gpurun_out
timeout 500 python -m pytest tests -m gpu -q -x > gpurun_out/c2_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/c2_tests.log
for k in 0 8 16; do
  echo "== HIPDEC_SEQ_LOOKAHEAD=$k"; HIPDEC_SEQ_LOOKAHEAD=$k timeout 300 python tools/sequence_fps.py 33 16 2>&1 | tail -4
done > gpurun_out/c2_seqfps.txt 2>&1
cat gpurun_out/c2_seqfps.txt
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print("%-12s" % sys.argv[1], d.get("value_resident", d["value"]), "ms/step", d["ms_per_step"], {k: round(v["avg_us"] / 1e3, 2) for k, v in d["kernels"].items()})
except Exception as e: print(sys.argv[1], "no line", e)
PY
}
for rep in 1 2; do
  timeout 200 python bench.py --only-main --steps 4 --warmup 1 > gpurun_out/c2_tree_$rep.json 2> gpurun_out/c2_tree_$rep.err; show tree/$rep gpurun_out/c2_tree_$rep.json
  HIPDEC_LIBRARY=$PWD/build/ab/candH/libheif_amd/libheifhip.so timeout 200 python bench.py --only-main --steps 4 --warmup 1 > gpurun_out/c2_candH_$rep.json 2> gpurun_out/c2_candH_$rep.err; show candH/$rep gpurun_out/c2_candH_$rep.json
  HIPDEC_PARSE_OCCUPANCY=6 HIPDEC_POOL_WAVES=6144 timeout 200 python bench.py --only-main --steps 4 --warmup 1 > gpurun_out/c2_occ6_$rep.json 2> gpurun_out/c2_occ6_$rep.err; show occ6/$rep gpurun_out/c2_occ6_$rep.json
done
