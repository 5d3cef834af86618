#!/bin/bash
# drop-in throughput at the given thread counts with the launch-set trace (dev tool)
# usage: bash tools/dropin_trace.sh "64,256,1024" [seconds] [trace lines] [extra args of dropin_throughput.py]
t=$1; sec=${2:-6}; n=${3:-8}; shift; shift; shift
python tools/dropin_throughput.py --threads "$t" --seconds $sec "$@" 2> gpurun_out/dropin_trace_$$.err | tail -4
grep "launch set" gpurun_out/dropin_trace_$$.err | tail -$n
