#!/bin/bash
# GPU probe of the lane-per-substream parser (parse kernel time per batch size; parity is tests/ with HIPDEC_PARSE_LANES=1)
mkdir -p gpurun_out
for n in ${LANES_PROBE_BATCHES:-2048}; do
  HIPDEC_PARSE_LANES=1 timeout 400 python bench.py --only-main --no-extras --no-cpu-baseline --steps 1 --warmup 1 --batch $n --distinct ${LANES_PROBE_DISTINCT:-64} > gpurun_out/lanes_bench_$n.json 2> gpurun_out/lanes_bench_$n.err
  echo "lanes n=$n rc=$?"; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/lanes_bench_$n.json').read().strip().split('\n')[-1])
print('value',d['value'],'ms/step',d['ms_per_step'],{k:round(v['avg_us']/1000,1) for k,v in d['kernels'].items()})"
done
