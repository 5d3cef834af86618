for cfg in "4 4096" "5 5120" "6 6144" "8 8192" "5 8192" "4 8192"; do
  set -- $cfg
  HIPDEC_PARSE_OCCUPANCY=$1 HIPDEC_POOL_WAVES=$2 python bench.py --only-main --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('occ $1 waves $2', d['value'], d['kernels']['parse']['avg_us'])"
done
