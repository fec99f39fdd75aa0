#!/bin/bash
# A/B of the per-triangle stage: the product build (triangle TU without the SLP vectoriser) against a variant with it
# (make -C renderer_amd/csrc variant W=5 TAG=_slp TRI_FLAGS=). usage: tools/r03_tri_ab.sh > gpurun_out/r03_tri_ab.txt
for rep in 1 2 3; do
  for lib in "" renderer_amd/lib/libmip_w5_slp.so; do
    for cfg in "2 100000" "3 200000" "2 20000" "2 1000"; do
      echo "== lib=${lib:-product (no SLP in the triangle TU)} cfg=$cfg rep=$rep"
      MIP_LIBRARY=$lib python3 tools/tri_bench.py $cfg 2>&1 | tail -1
    done
  done
done
