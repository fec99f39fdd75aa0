#!/bin/bash
# Round 5: the range kernel (equal ranges of the triangle stream, one or a few per wave) against the round-4 kernels
# (MIP_TUNE_TRI_CHUNKS_FROM=4294967295). usage: tools/r05_tri_ranges.sh [reps] [slots per range values...]
REPS=${1:-2}
shift
RPW="${@:-4096}"
for rep in $(seq 1 $REPS); do
  for cfg in "2 1000" "2 5000" "2 20000" "2 100000" "2 100000 strips" "2 100000 shuffled" "2 300000" "3 1000" "3 20000" "3 100000" "3 200000" "3 1000000"; do
    echo "== r04 kernels cfg=$cfg rep=$rep"
    MIP_TUNE_TRI_CHUNKS_FROM=4294967295 python3 tools/tri_bench.py $cfg 2>&1 | tail -1
    for r in $RPW; do
      echo "== ranges of $r cfg=$cfg rep=$rep"
      MIP_TUNE_TRI_RANGE_SLOTS=$r python3 tools/tri_bench.py $cfg 2>&1 | tail -1
    done
  done
done
