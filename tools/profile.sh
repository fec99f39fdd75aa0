#!/bin/bash
# Runs on the GPU box: kernel-trace stats for the default bench command, then PMC passes.
# usage: tools/profile.sh <tag>   -> gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -- python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra $BENCH_ARGS > $ROOT/$OUT/bench_under_rocprof.json 2> $ROOT/$OUT/trace.log
cd $ROOT
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace_full.csv
head -20 $OUT/kernel_stats.csv
