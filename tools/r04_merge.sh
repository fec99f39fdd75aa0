#!/bin/bash
# round 4: the wire merge — timing on the 8-rank / 2-rank shapes of BASELINE configs[3] (tools/merge_bench.py: lone launches
# and back to back), and the kernels' durations in a rocprofv3 kernel trace
mkdir -p gpurun_out
O=gpurun_out/r04_wire_merge_final.txt
: > $O
for form in packed wire cmds; do for R in 8 2; do timeout -k 10 120 python tools/merge_bench.py $R $form 2>&1 | grep -v amdgpu.ids >> $O || exit $?; done; done
bash tools/r04_merge_prof.sh >> $O 2>&1
cat $O
