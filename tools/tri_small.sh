# small-frame per-triangle stage: parts kernel (default) against the workgroup-per-command kernels
for n in 30 200 1000 2000 4000; do
  echo "== n=$n parts"; python tools/tri_bench.py 2 $n 2>&1 | grep frame
  echo "== n=$n blocks"; MIP_TUNE_TRI_PARTS_MAX=0 python tools/tri_bench.py 2 $n 2>&1 | grep frame
done
echo "== mixed scene n=3000 parts"; python tools/tri_bench.py 3 3000 2>&1 | grep frame
echo "== mixed scene n=3000 blocks"; MIP_TUNE_TRI_PARTS_MAX=0 python tools/tri_bench.py 3 3000 2>&1 | grep frame
