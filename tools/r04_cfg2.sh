#!/bin/bash
# round 4, BASELINE configs[1] (100 k, one mesh) and its neighbours: current build / round 3 / tuning variants, same box, interleaved
mkdir -p gpurun_out
LIBS=${LIBS:-default,renderer_amd/lib/libmip_r03.so,renderer_amd/lib/libmip_w5_early.so,default,renderer_amd/lib/libmip_r03.so,renderer_amd/lib/libmip_w5_early.so}
timeout -k 10 500 python tools/kbench.py --configs 2,2,3,3,3 --n 100000,30000,100000,300000,700000 --libs $LIBS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_cfg2_ab.txt
