#!/bin/bash
# tuning: where the wire merge's time goes — variants built with make variant EXTRA=-DMIP_MERGE_EXP=.. / -DMIP_MERGE_STORE_NT=..
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_merge_exp
mkdir -p $OUT
cd /tmp
for lib in ${LIBS:-default libmip_w5_mnostore.so libmip_w5_mnotable.so libmip_w5_msc1nt.so libmip_w5_mnt.so}; do
  for form in packed wire; do
    if [ $lib = default ]; then unset MIP_LIBRARY; else export MIP_LIBRARY=$ROOT/renderer_amd/lib/$lib; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$lib.$form -- python3 $ROOT/tools/merge_bench.py 8 $form > $OUT/$lib.$form.log 2>&1 || { tail -5 $OUT/$lib.$form.log; exit 1; }
    f=$(find $OUT/$lib.$form -name "*kernel_stats.csv" | head -1)
    echo "$lib $form: $(grep -i 'merge' $f | cut -d, -f1-4,7-8 | cut -c1-160)"
  done
done
