#!/bin/bash
# ordered-tiles mode by launch size: one ticket per tile against the three wait-free launches, against the default kernel
for n in 100000 262144 400000 1000000 4000000 10000000; do
  cfg=3; [ $n -ge 4000000 ] && cfg=4
  echo "== n=$n"
  echo -n "default            "; python tools/kbench.py --configs $cfg --n $n 2>&1 | grep -v amdgpu.ids | tail -1
  echo -n "ordered, tickets   "; MIP_TUNE_ORDERED_TILES=1 MIP_TUNE_THREE_PASS_MIN_TILES=100000000 python tools/kbench.py --configs $cfg --n $n 2>&1 | grep -v amdgpu.ids | tail -1
  echo -n "ordered, three-pass"; MIP_TUNE_ORDERED_TILES=1 MIP_TUNE_THREE_PASS_MIN_TILES=0 python tools/kbench.py --configs $cfg --n $n 2>&1 | grep -v amdgpu.ids | tail -1
done
