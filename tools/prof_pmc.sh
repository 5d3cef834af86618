# usage: bash tools/prof_pmc.sh <tag> "<counters>" [bench args...]
tag=$1; shift; ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o $tag -- python $GRAFT_REPO_ROOT/bench.py "$@" > $out/bench.json 2> $out/bench.err
echo "rc=$?"
ls $out; f=$(find $out -name '*counter_collection.csv' | head -1)
python - <<PY
import csv, collections
f="$f"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:40]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    n[(k,r["Counter_Name"])]+=1
for k in agg:
    print(k)
    for c,v in agg[k].items(): print("   %-28s total %.4g  per-dispatch %.4g (n=%d)"%(c,v,v/n[(k,c)],n[(k,c)]))
PY
