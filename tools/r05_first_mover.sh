#!/bin/bash
# Round 5: the frame kernel whose tiles mark themselves STARTED and whose helpers add to the group accumulators (product) against
# the rule of rounds 3-4 (only owners add, nothing marks a start: make -C renderer_amd/csrc variant W=5 TAG=_ownersonly
# EXTRA=-DMIP_FIRST_MOVER_ADDS=0) in the ORDINARY dispatch order: what the swap at the head of every tile costs. Same box, interleaved.
mkdir -p gpurun_out/r05
LIBS=${LIBS:-default,renderer_amd/lib/libmip_w5_ownersonly.so,default,renderer_amd/lib/libmip_w5_ownersonly.so,default,renderer_amd/lib/libmip_w5_ownersonly.so}
timeout -k 10 600 python tools/kbench.py --configs 2,3,3,3,4 --n 100000,300000,1000000,2500000,10000000 --libs $LIBS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/first_mover_ab.txt
