cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_golden.py -q -m gpu -x 2>&1 | tail -2
bash tools/_pmc.sh g "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" --steps 1 --warmup 0 --batch 512 --streams 1 --no-cpu-baseline 2>&1 | grep -A 9 "k_recon\|k_sao\|k_residual" | head -34
python -c "
import json; d=json.load(open('gpurun_out/pmc_g/bench.json')); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
