#!/bin/bash
# SQ / GRBM counters of the instance kernel (where its wave-cycles go). usage: tools/pmc_sq.sh <tag> <config>
# One counter group per rocprofv3 run, kernel-trace only (no other tracing domain beside --pmc).
TAG=${1:-r02}; CFG=${2:-3}
OUT=$(pwd)/gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/pmc_run.py full $CFG 20 > $OUT/$name.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json, hashlib
out="$OUT"
acc=collections.defaultdict(list)
dur=[]
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mip_instance_pipeline" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{out}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mip_instance_pipeline" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m={k: sum(v)/len(v) for k,v in acc.items()}
sha=hashlib.sha256()
for f in ("instance_kernel.hpp",): sha.update(open("$ROOT/renderer_amd/csrc/"+f,"rb").read())
doc=dict(kernel_source_sha=sha.hexdigest()[:16], config=$CFG, launches_per_pass=20, counters=m,
         kernel_ns_under_pmc=(sum(dur)/len(dur) if dur else None),
         note="means per launch over all passes; SQ_*_CYCLES and SQ_WAIT_*/SQ_ACTIVE_* count quad-cycles summed over waves (MI355X_MICROARCH.md, rocprofv3 PMC slots); WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES")
if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
    wc=m["SQ_WAVE_CYCLES"]
    doc["fractions_of_wave_cycles"]={k: m[k]/wc for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_WAIT_INST_LDS") if k in m}
if m.get("SQ_WAVE_CYCLES") and m.get("GRBM_GUI_ACTIVE"):
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_WAVE_CYCLES counts quad-cycles summed over waves
    doc["mean_resident_waves_per_simd"]=m["SQ_WAVE_CYCLES"]*4.0/((m["GRBM_GUI_ACTIVE"]/8.0)*1024.0)
if "SQ_WAVES" in m and m.get("SQ_WAVES"):
    doc["valu_instructions_per_wave"]=m.get("SQ_INSTS_VALU",0)/m["SQ_WAVES"]
json.dump(doc, open(f"{out}/sq_summary.json","w"), indent=1)
print(json.dumps(doc, indent=1))
PY
