#!/bin/bash
# kernel durations of the wire merge (8-rank shape) under rocprofv3 --kernel-trace: what the step pays when the merge
# is queued behind the all-gather on one stream (merge_bench.py's hipEvent pair around a lone launch adds the launch)
set -o pipefail
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_merge_prof
mkdir -p $OUT
cd /tmp
for form in packed wire cmds; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$form -- python3 $ROOT/tools/merge_bench.py 8 $form > $OUT/$form.log 2>&1 || { tail -5 $OUT/$form.log; exit 1; }
  f=$(find $OUT/$form -name "*kernel_stats.csv" | head -1)
  echo "== $form"; grep -i "merge" $f | cut -c1-200
done
