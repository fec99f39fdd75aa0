#!/bin/bash
# Round 4 closing run: the whole GPU suite, then the default bench line. usage: tools/r04_final.sh (under gpurun)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x > gpurun_out/r04_final_gpu_tests.txt 2>&1 || { tail -30 gpurun_out/r04_final_gpu_tests.txt; exit 1; }
tail -3 gpurun_out/r04_final_gpu_tests.txt
timeout -k 10 600 python3 bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err || { tail -20 gpurun_out/r04_final_bench.err; exit 1; }
tail -c 600 gpurun_out/r04_final_bench.json
