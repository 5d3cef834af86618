mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -q > gpurun_out/final_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/final_tests.log
timeout 700 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; tail -c 400 gpurun_out/final_bench.err; head -c 600 gpurun_out/final_bench.json; echo
