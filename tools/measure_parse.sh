bash tools/gpu_stage.sh $1 benchmain; cd /tmp && export TMPDIR=/tmp; export HIPDEC_SYNC_UPLOAD=1; timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1_pmc -o p -- python $GRAFT_REPO_ROOT/bench.py --only-main --steps 1 --warmup 0 --batch 512 > /dev/null 2>&1; python - <<PY
import csv,collections,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$1_pmc/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "k_parse" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
px=512*3840*2160
print({k: round(v/px,3) for k,v in agg.items()})
PY
