#!/bin/bash
# A/B of the per-triangle stage. usage: tools/r03_tri_ab.sh "<lib A> <lib B> ..." ("-" = the product build) [reps] > gpurun_out/....txt
LIBS=${1:-"- renderer_amd/lib/libmip_w5_slp.so"}
REPS=${2:-3}
for rep in $(seq 1 $REPS); do
  for lib in $LIBS; do
    [ "$lib" = "-" ] && lib=""
    for cfg in "2 100000" "2 100000 strips" "2 100000 shuffled" "3 200000" "2 20000" "2 1000"; do
      echo "== lib=${lib:-product} cfg=$cfg rep=$rep"
      MIP_LIBRARY=$lib python3 tools/tri_bench.py $cfg 2>&1 | tail -1
    done
  done
done
