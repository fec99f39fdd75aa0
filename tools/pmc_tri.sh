#!/bin/bash
# SQ / cache / traffic counters of the per-triangle kernels (row f-1). usage: tools/pmc_tri.sh <tag> [config] [n]
# One counter group per rocprofv3 run, kernel-trace only beside --pmc.
TAG=${1:-tri}; CFG=${2:-2}; N=${3:-100000}; ORDERING=${4:-rows}
OUT=$(pwd)/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
[ -n "$MIP_LIBRARY" ] && export MIP_LIBRARY=$(realpath $MIP_LIBRARY)
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/tri_bench.py $CFG $N $ORDERING > $OUT/$name.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json, hashlib
out="$OUT"
acc=collections.defaultdict(list); dur=collections.defaultdict(list)
# large frames launch two grids and one returns at once (tri_choice_is_block): the counters of the one that did the work
KERNEL = "${PMC_TRI_KERNEL:-mip_triangle_cull_kernel(}"
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{out}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m={k: sum(v)/len(v) for k,v in acc.items()}
sha=hashlib.sha256(open("$ROOT/renderer_amd/csrc/triangle_kernels.hpp","rb").read()).hexdigest()[:16]
kern={k: sum(v)/len(v) for k,v in dur.items()}
doc=dict(triangle_source_sha=sha, config=$CFG, instances=$N, ordering="$ORDERING", library="${MIP_LIBRARY:-product}", counters_per_launch=m, kernel_ns_under_pmc=kern,
         note="means per launch of the wave-per-command kernel (PMC_TRI_KERNEL selects another) over the 13 frames tri_bench.py runs; SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are quad-cycles summed over waves")
if m.get("SQ_WAVE_CYCLES"):
    doc["fractions_of_wave_cycles"]={k: m[k]/m["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU") if k in m}
if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum",0)+m.get("TCC_MISS_sum",0)):
    doc["l2_hit_rate"]=m["TCC_HIT_sum"]/(m["TCC_HIT_sum"]+m["TCC_MISS_sum"])
json.dump(doc, open(f"{out}/tri_pmc_summary.json","w"), indent=1)
print(json.dumps(doc, indent=1))
PY
