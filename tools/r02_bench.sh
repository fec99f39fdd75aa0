# bench line (default), the N>1 path rehearsed with one rank, rocprofv3 stats + PMC passes for BASELINE config 3
set -o pipefail
mkdir -p gpurun_out/r02
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r02/bench_default.json
MIP_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 10 --warmup 5 > gpurun_out/r02/bench_dist1.json 2> gpurun_out/r02/bench_dist1.err; echo "dist rc=$?"; tail -c 2500 gpurun_out/r02/bench_dist1.json; tail -5 gpurun_out/r02/bench_dist1.err
