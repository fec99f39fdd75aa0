for cfg in "2 70000" "2 80000" "2 150000" "3 600000" "3 1000000"; do
  for ch in block waves; do
    echo "== cfg=$cfg choice=$ch"; MIP_TUNE_TRI_CHOICE=$ch python3 tools/tri_bench.py $cfg 2>&1 | tail -1
  done
done
