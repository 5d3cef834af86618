# usage: bash tools/prof_kernel_stats.sh <tag> [bench args...]; leaves rocprofv3 stats CSVs under gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python $GRAFT_REPO_ROOT/bench.py "$@" > $out/bench.json 2> $out/bench.err
echo "rc=$?"
find $out -name '*stats*' | head; cat $out/bench.json
f=$(find $out -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -20 $f
