cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_golden.py -q -m gpu -x 2>&1 | tail -3
bash tools/_prof.sh r1g --steps 2 --warmup 1 --no-cpu-baseline --batch 512 --streams 1 | grep -i "k_re\|k_sao\|k_parse_occ\|value" | cut -c1-150
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; python -c "
import json; d=json.load(open('gpurun_out/bench_x.json')); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
