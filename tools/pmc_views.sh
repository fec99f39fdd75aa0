#!/bin/bash
# SQ / traffic counters of the multi-view kernel (row f-4, four views of 1 M instances). usage: tools/pmc_views.sh <tag> [n]
# One counter group per rocprofv3 run, kernel-trace only beside --pmc.
TAG=${1:-views}; N=${2:-1000000}
OUT=$(pwd)/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
[ -n "$MIP_LIBRARY" ] && export MIP_LIBRARY=$(realpath $MIP_LIBRARY)
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/views_pmc_run.py $N 12 > $OUT/$name.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json, hashlib
out="$OUT"
acc=collections.defaultdict(list); dur=[]
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cull_views" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{out}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cull_views" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m={k: sum(v)/len(v) for k,v in acc.items()}
sha=hashlib.sha256(open("$ROOT/renderer_amd/csrc/views_kernel.hpp","rb").read()).hexdigest()[:16]
doc=dict(views_source_sha=sha, instances=$N, views=4, library="${MIP_LIBRARY:-product}", counters_per_launch=m,
         kernel_ns_under_pmc=sum(dur)/max(len(dur),1), launches=len(dur),
         note="means per launch of mip_cull_views_kernel; SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are quad-cycles summed over waves; "
              "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; hbm_bytes_per_launch = 2 * FETCH + WRITE (gfx950 reports half of a wide "
              "coalesced read stream, MI355X_MICROARCH.md HBM section, as tools/pmc.sh)")
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    doc["hbm_bytes_per_launch"]=2*m["FETCH_SIZE"]*1024+m["WRITE_SIZE"]*1024
if m.get("SQ_WAVE_CYCLES"):
    doc["fractions_of_wave_cycles"]={k: m[k]/m["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU") if k in m}
json.dump(doc, open(f"{out}/views_pmc_summary.json","w"), indent=1)
print(json.dumps(doc, indent=1))
PY
