cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/debug_decode.py 2>&1 | tail -12
timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_golden.py -q -m gpu -x 2>&1 | tail -12
