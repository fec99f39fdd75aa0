# same-box A/B of library variants: bash tools/r02_ab.sh <tag> <libs comma list> ; env sweeps appended by hand
set -o pipefail
tag=$1; libs=$2
mkdir -p gpurun_out/r02
out=gpurun_out/r02/ab_$tag.txt
: > $out
for rep in 1 2; do python tools/kbench.py --configs 3 --libs $libs >> $out 2>&1; done
python tools/kbench.py --configs 2,4 --n ,10000000 --libs $libs >> $out 2>&1
for pad in 1000 7000 13000; do echo "LDS pad $pad" >> $out; MIP_TUNE_LDS_PAD=$pad python tools/kbench.py --configs 3 >> $out 2>&1; done
grep -v amdgpu.ids $out
