#!/bin/bash
# Round 5: where the range kernel's time goes against the wave-per-command kernel: kernel durations (rocprofv3 kernel trace) of one-mesh frames.
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05_tri_probe
mkdir -p $OUT
cd /tmp
for mode in r04 ranges; do
  for cfg in "2 100000" "2 300000"; do
    tag=${mode}_$(echo $cfg | tr ' ' '_')
    if [ $mode = r04 ]; then export MIP_TUNE_TRI_CHUNKS_FROM=4294967295; else unset MIP_TUNE_TRI_CHUNKS_FROM; fi
    MIP_TUNE_VERBOSE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $ROOT/tools/tri_bench.py $cfg > $OUT/$tag.log 2>&1
    echo "== $tag"; grep "mip:" $OUT/$tag.log | head -2
    f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1); head -8 $f | cut -d, -f1-6
  done
done
unset MIP_TUNE_TRI_CHUNKS_FROM
cd $ROOT
for per in 4 6 8; do echo "== ranges, $per workgroups per CU, cfg 2 300000"; MIP_TUNE_TRI_RANGE_BLOCKS_PER_CU=$per python3 tools/tri_bench.py 2 300000 | tail -1; done
