#!/bin/bash
# PMC passes for the triangle kernel. usage: tools/pmc_tri.sh <tag>
TAG=${1:-tri}
OUT=$(pwd)/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/tri_bench.py 2 20000 > $OUT/$name.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
out="$OUT"
acc=collections.defaultdict(list)
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "triangle_cull" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    print(f"{k:36s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
