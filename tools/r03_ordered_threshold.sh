for n in 65536 100000 131072 160000 200000; do
  echo "== n=$n"
  echo -n "tickets "; MIP_TUNE_ORDERED_TILES=1 MIP_TUNE_THREE_PASS_MIN_TILES=100000000 python tools/kbench.py --configs 3 --n $n 2>&1 | grep -v amdgpu.ids | tail -1
  echo -n "three   "; MIP_TUNE_ORDERED_TILES=1 MIP_TUNE_THREE_PASS_MIN_TILES=0 python tools/kbench.py --configs 3 --n $n 2>&1 | grep -v amdgpu.ids | tail -1
done
