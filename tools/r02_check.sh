# usage: bash tools/r02_check.sh <tag> [extra kbench args]   — parity tests + serialized timing + stamps for the built library
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gputest_$tag.txt 2>&1; tail -3 gpurun_out/r02/gputest_$tag.txt
python tools/kbench.py --configs 3,2,4 --n ,,10000000 "$@" > gpurun_out/r02/kbench_$tag.txt 2>&1; grep -v amdgpu.ids gpurun_out/r02/kbench_$tag.txt
python tools/stamps.py 3 > gpurun_out/r02/stamps_$tag.txt 2>&1; grep -v amdgpu.ids gpurun_out/r02/stamps_$tag.txt
