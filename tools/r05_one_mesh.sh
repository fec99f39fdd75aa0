#!/bin/bash
# Round 5: one-entry mesh tables (BASELINE configs[1]) — the entry as a scalar load beside the instance loads (product) against the id
# load + gather every other table takes (MIP_TUNE_NO_ONE_MESH=1, read at context creation). Same box, same library, interleaved.
mkdir -p gpurun_out/r05
for round in 1 2 3; do
  for v in product gather; do
    if [ $v = gather ]; then export MIP_TUNE_NO_ONE_MESH=1; else unset MIP_TUNE_NO_ONE_MESH; fi
    timeout -k 10 300 python tools/kbench.py --configs 2,2,2,2 --n 30000,100000,300000,1000000 --libs default 2>&1 | grep -v amdgpu.ids | sed "s/^default  /$v/"
  done
done | tee gpurun_out/r05/one_mesh_ab.txt
