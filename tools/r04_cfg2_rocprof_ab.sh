#!/bin/bash
# kernel durations of the 100 k frame (BASELINE configs[1]) in a rocprofv3 kernel trace: this build against round 3's library, same box
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_cfg2_rocprof_ab
mkdir -p $OUT
cd /tmp
for lib in default libmip_r03.so default libmip_r03.so; do
  if [ $lib = default ]; then unset MIP_LIBRARY; else export MIP_LIBRARY=$ROOT/renderer_amd/lib/$lib; fi
  rm -rf $OUT/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$lib -- python3 $ROOT/tools/kbench.py --child --configs 2,3 --n 100000,300000 > $OUT/$lib.log 2>&1 || { tail -5 $OUT/$lib.log; exit 1; }
  f=$(find $OUT/$lib -name "*kernel_stats.csv" | head -1)
  grep -v amdgpu.ids $OUT/$lib.log | tail -2
  echo "$lib: $(grep 'mip_instance' $f | cut -d, -f1-4,6-8 | cut -c1-200)"
done
