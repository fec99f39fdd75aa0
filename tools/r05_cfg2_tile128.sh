#!/bin/bash
# Round 5, BASELINE configs[1] (100 k, one mesh) and its neighbours: the product (256 instances per tile = workgroup) against an
# experiment build of the WHOLE library with -DMIP_TILE=128 (make -C renderer_amd/csrc variant W=5 TAG=_tile128 EXTRA=-DMIP_TILE=128:
# 782 workgroups of two waves instead of 391 of four at 100 k; the multi-view kernel is not usable in that build). Same box, interleaved.
mkdir -p gpurun_out
LIBS=${LIBS:-default,renderer_amd/lib/libmip_w5_tile128.so,default,renderer_amd/lib/libmip_w5_tile128.so}
timeout -k 10 500 python tools/kbench.py --configs 2,2,2,3,3,3 --n 100000,30000,200000,100000,300000,1000000 --libs $LIBS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_cfg2_tile128_ab.txt
