#!/bin/bash
# Round 5: kernel by kernel what a per-triangle frame costs (rocprofv3 kernel trace of tools/tri_bench.py). usage: tools/r05_tri_probe.sh "<cfg n>" ...
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05_tri_probe
mkdir -p $OUT
cd /tmp
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $ROOT/tools/tri_bench.py $cfg > $OUT/$tag.log 2>&1
  echo "== $cfg"; tail -1 $OUT/$tag.log
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1); head -12 $f | cut -d, -f1-6
done
