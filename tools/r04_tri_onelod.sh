#!/bin/bash
# Round 4: what bounds the per-triangle kernel? The same frame with every command on LOD 0 (TRI_BENCH_ONE_LOD=1:
# consecutive triangles use consecutive vertices) against the mixed-LOD frame (the synthetic LOD 1 is a strided subset
# of LOD 0's triangles: a gather touches ~3x the lines). profiles/r04_triangle_bound_experiments.txt reads the result.
for cfg in "2 100000" "2 100000 strips"; do
  for one in 1 0; do
    echo "== one_lod=$one cfg=$cfg"
    TRI_BENCH_ONE_LOD=$one python3 tools/tri_bench.py $cfg 2>&1 | tail -1
  done
done
