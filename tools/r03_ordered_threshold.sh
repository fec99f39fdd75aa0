#!/bin/bash
# ordered-tiles mode on small launches: one ticket per tile against the wait-free launches (frame without commands + commands from the bitmap;
# no scan launch at these sizes), forced onto every size
for n in 4096 16384 32768 65536 100000 131072; do
  echo "== n=$n"
  echo -n "tickets "; MIP_TUNE_ORDERED_TILES=1 MIP_TUNE_THREE_PASS_MIN_TILES=100000000 python tools/kbench.py --configs 3 --n $n 2>&1 | grep -v amdgpu.ids | tail -1
  echo -n "two     "; MIP_TUNE_ORDERED_TILES=1 MIP_TUNE_THREE_PASS_MIN_TILES=0 python tools/kbench.py --configs 3 --n $n 2>&1 | grep -v amdgpu.ids | tail -1
  echo -n "default "; python tools/kbench.py --configs 3 --n $n 2>&1 | grep -v amdgpu.ids | tail -1
done
