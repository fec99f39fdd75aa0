#!/bin/bash
# Round 5: cache policy of the culled index stream's stores (MIP_TRI_STORE_AUX: 0 plain, 2 nt, 18 sc1 nt) and the range size, same box.
for rep in 1 2; do
  for cfg in "2 100000" "3 100000" "3 1000000" "2 100000 shuffled"; do
    for lib in default libmip_w5_nt.so libmip_w5_sc1nt.so; do
      echo "== $lib cfg=$cfg rep=$rep"
      if [ $lib = default ]; then python3 tools/tri_bench.py $cfg 2>&1 | tail -1; else MIP_LIBRARY=$PWD/renderer_amd/lib/$lib python3 tools/tri_bench.py $cfg 2>&1 | tail -1; fi
    done
  done
done
for slots in 2048 3072 4096 6144 8192; do echo "== range slots $slots cfg=3 100000"; MIP_TUNE_TRI_RANGE_SLOTS=$slots python3 tools/tri_bench.py 3 100000 2>&1 | tail -1; echo "== range slots $slots cfg=3 20000"; MIP_TUNE_TRI_RANGE_SLOTS=$slots python3 tools/tri_bench.py 3 20000 2>&1 | tail -1; done
