#!/bin/bash
# ordered-tiles mode by launch size: tickets (small launches' way, forced here for every size) against the three wait-free launches
# (frame without commands + group scan + commands from the bitmap), against the default kernel
for n in 262144 1000000 4000000 10000000; do
  cfg=3; [ $n -ge 4000000 ] && cfg=4
  echo "== n=$n"
  echo -n "default kernel        "; python tools/kbench.py --configs $cfg --n $n 2>&1 | grep -v amdgpu.ids | tail -1
  echo -n "ordered, three launches"; MIP_TUNE_ORDERED_TILES=1 python tools/kbench.py --configs $cfg --n $n 2>&1 | grep -v amdgpu.ids | tail -1
  echo -n "ordered, tickets      "; MIP_TUNE_ORDERED_TILES=1 MIP_TUNE_THREE_PASS_MIN_TILES=100000000 python tools/kbench.py --configs $cfg --n $n 2>&1 | grep -v amdgpu.ids | tail -1
done
