# dev tool: sweep of the CABAC work-pool knobs on the default bench workload (GPU box)
run() { echo -n "$* : "; env "$@" timeout 300 python bench.py --streams ${STREAMS:-1} --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], round(d['kernels']['parse']['avg_us']/1e3,1))"; }
for cfg in "$@"; do run $cfg; done
