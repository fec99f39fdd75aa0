#!/bin/bash
# round 4, first GPU call: order-independence of the self-helping look-up + same-box A/B against round 3's kernel
mkdir -p gpurun_out
timeout -k 10 700 python tools/r04_selfhelp_check.py > gpurun_out/r04_selfhelp.txt 2>&1
rc=$?
echo "selfhelp rc=$rc" | tee -a gpurun_out/r04_selfhelp.txt
tail -5 gpurun_out/r04_selfhelp.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python tools/kbench.py --configs 3,2 --libs default,renderer_amd/lib/libmip_r03.so,default,renderer_amd/lib/libmip_r03.so > gpurun_out/r04_kbench_ab.txt 2>&1 || exit $?
timeout -k 10 300 python tools/kbench.py --configs 3,3,3 --n 10000000,300000,4000000 --libs default,renderer_amd/lib/libmip_r03.so >> gpurun_out/r04_kbench_ab.txt 2>&1
cat gpurun_out/r04_kbench_ab.txt
