#!/bin/bash
# same-box A/B of the serialized frame time: the current build against round 3's kernel (renderer_amd/lib/libmip_r03.so)
mkdir -p gpurun_out
LIBS=${LIBS:-default,renderer_amd/lib/libmip_r03.so,default,renderer_amd/lib/libmip_r03.so}
timeout -k 10 400 python tools/kbench.py --configs 3,2 --libs $LIBS > gpurun_out/r04_kbench_ab.txt 2>&1 || exit $?
timeout -k 10 400 python tools/kbench.py --configs 3,3,3 --n 10000000,300000,4000000 --libs $LIBS >> gpurun_out/r04_kbench_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04_kbench_ab.txt
