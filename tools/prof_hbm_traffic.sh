# HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X guide: TCC slots)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$c
mkdir -p $out
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 256 --streams 1 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
f=$(find $out -name '*counter_collection.csv' | head -1)
python - <<PY
import csv, collections
agg=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open("$f")):
    k=r["Kernel_Name"].split("(")[0][:40]; agg[k]+=float(r["Counter_Value"]); n[k]+=1
for k in agg: print("$c %-44s total %.6g (dispatches %d)"%(k,agg[k],n[k]))
PY
done
