#!/bin/bash
# A/B of the whole library built with / without the SLP vectoriser (make variant W=5 TAG=_noslp EXTRA=-fno-slp-vectorize)
mkdir -p gpurun_out/r03
L=$(pwd)/renderer_amd/lib/libmip_w5_noslp.so
for rep in 1 2; do
  for lib in default $L; do
    if [ $lib = default ]; then unset MIP_LIBRARY; else export MIP_LIBRARY=$lib; fi
    echo "== $lib"
    python tools/light_bench.py 2>&1 | grep -v amdgpu.ids
    python tools/skin_bench.py 2>&1 | grep -v amdgpu.ids
    python tools/merge_bench.py 8 2>&1 | grep -v amdgpu.ids
  done
done
unset MIP_LIBRARY
python tools/kbench.py --configs 3,3,3,3,4 --n 200000,500000,800000,4000000,10000000 --libs default,$L,default,$L 2>&1 | grep -v amdgpu.ids
