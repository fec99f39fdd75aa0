#!/bin/bash
# Round 4: large per-triangle frames launch the wave-per-command AND the workgroup-per-command grid and pick one on the device
# (tri_choice_is_block) against always the wave-per-command kernel (MIP_TUNE_TRI_NO_CHOICE=1). usage: tools/r04_tri_choice.sh [reps]
REPS=${1:-2}
for rep in $(seq 1 $REPS); do
  for cfg in "2 100000" "2 100000 strips" "2 100000 shuffled" "2 300000" "3 70000" "3 100000" "3 200000" "3 400000" "3 1000000"; do
    for mode in choice waves; do
      echo "== mode=$mode cfg=$cfg rep=$rep"
      if [ $mode = waves ]; then MIP_TUNE_TRI_NO_CHOICE=1 python3 tools/tri_bench.py $cfg 2>&1 | tail -1; else python3 tools/tri_bench.py $cfg 2>&1 | tail -1; fi
    done
  done
done
