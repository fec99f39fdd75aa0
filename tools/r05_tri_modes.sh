#!/bin/bash
# Round 5: the per-triangle stage's kernels against each other. r04 = the round-4 kernels (MIP_TUNE_TRI_CHUNKS_FROM=4294967295);
# default = range kernel up to 65 536 instances, above: range kernel OR wave-per-command over size-sorted commands, chosen on the device;
# ranges / waves = that choice forced (MIP_TUNE_TRI_CHOICE). usage: tools/r05_tri_modes.sh [reps] [cfg ...]
REPS=${1:-2}
shift
if [ $# -gt 0 ]; then CFGS=("$@"); else CFGS=("2 1000" "2 5000" "2 20000" "2 70000" "2 100000" "2 100000 strips" "2 100000 shuffled" "2 300000" "3 1000" "3 20000" "3 70000" "3 100000" "3 200000" "3 400000" "3 1000000"); fi
for rep in $(seq 1 $REPS); do
  for cfg in "${CFGS[@]}"; do
    echo "== r04 cfg=$cfg rep=$rep";     MIP_TUNE_TRI_CHUNKS_FROM=4294967295 python3 tools/tri_bench.py $cfg 2>&1 | tail -1
    echo "== default cfg=$cfg rep=$rep"; python3 tools/tri_bench.py $cfg 2>&1 | tail -1
    n=$(echo $cfg | cut -d' ' -f2)
    if [ $n -gt 65536 ]; then
      echo "== ranges cfg=$cfg rep=$rep"; MIP_TUNE_TRI_CHOICE=block python3 tools/tri_bench.py $cfg 2>&1 | tail -1
      echo "== waves cfg=$cfg rep=$rep";  MIP_TUNE_TRI_CHOICE=waves python3 tools/tri_bench.py $cfg 2>&1 | tail -1
    fi
  done
done
