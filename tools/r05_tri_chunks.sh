#!/bin/bash
# Round 5: the chunk kernel (equal chunks of the triangle stream, one wave each) against the round-4 kernels
# (MIP_TUNE_TRI_CHUNKS_FROM=4294967295) and against build variants (MIP_LIBRARY). usage: tools/r05_tri_chunks.sh [reps] [libs...]
REPS=${1:-2}
shift
LIBS="$@"
for rep in $(seq 1 $REPS); do
  for cfg in "2 1000" "2 20000" "2 100000" "2 100000 strips" "2 100000 shuffled" "2 300000" "3 1000" "3 20000" "3 100000" "3 200000" "3 1000000"; do
    echo "== r04 kernels cfg=$cfg rep=$rep"
    MIP_TUNE_TRI_CHUNKS_FROM=4294967295 python3 tools/tri_bench.py $cfg 2>&1 | tail -1
    echo "== chunks cfg=$cfg rep=$rep"
    python3 tools/tri_bench.py $cfg 2>&1 | tail -1
    for lib in $LIBS; do
      echo "== chunks $lib cfg=$cfg rep=$rep"
      MIP_LIBRARY=$PWD/renderer_amd/lib/$lib python3 tools/tri_bench.py $cfg 2>&1 | tail -1
    done
  done
done
