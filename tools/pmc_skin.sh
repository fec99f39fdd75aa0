#!/bin/bash
# Kernel trace + PMC passes of a skinned frame (BASELINE config 5: mip_skinned_bounds_kernel + the frame kernel with a box override).
# usage: tools/pmc_skin.sh <tag>   -> gpurun_out/pmc_<tag>/skinned_pmc_summary.json (+ kernel_stats_{palette,bounds}.csv)
# One counter group per rocprofv3 run, kernel-trace only beside --pmc. The PMC passes run the frame WITH the palette (what
# bench.py's skinned_256k leg runs); the two kernel traces keep the variants apart (round 3's stats file mixed them).
TAG=${1:-skin}
OUT=$(pwd)/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
for v in palette bounds; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $ROOT/tools/skin_bench.py 256000 40 $v > $OUT/trace_$v.log 2>&1
  find $OUT/trace_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$v.csv
done
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/skin_bench.py 256000 8 palette > $OUT/$name.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json, hashlib
out="$OUT"
kinds={"skinned": "mip_skinned_bounds_kernel", "mip_instance": "mip_instance_pipeline_kernel"}
acc={k: collections.defaultdict(list) for k in kinds}
dur={k: [] for k in kinds}
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in kinds:
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{out}/trace_palette/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in kinds:
            if k in r["Kernel_Name"]:
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
h=hashlib.sha256()
for f in ("skinning_kernel.hpp", "instance_kernel.hpp"):
    h.update(open("$ROOT/renderer_amd/csrc/"+f,"rb").read())
per={}
total=0.0
for k, name in kinds.items():
    m={c: sum(v)/len(v) for c, v in acc[k].items()}
    row=dict(counters_per_launch=m, kernel_ns=(sum(dur[k])/len(dur[k]) if dur[k] else None), launches_traced=len(dur[k]))
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        row["hbm_bytes_per_launch"]=2*m["FETCH_SIZE"]*1024+m["WRITE_SIZE"]*1024
        total+=row["hbm_bytes_per_launch"]
    per[name]=row
doc=dict(source_sha=h.hexdigest()[:16], instances=256000, joints=19, variant="with palette", per_kernel=per, hbm_bytes_per_frame=total or None,
         note="means per launch; a frame = one launch of each kernel; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, hbm bytes = 2 * FETCH + WRITE "
              "(gfx950 reports half of a wide coalesced read stream, MI355X_MICROARCH.md HBM section, as tools/pmc.sh); kernel_ns from the un-instrumented kernel trace "
              "of the same variant; SQ_* cycle counters are quad-cycles summed over waves")
json.dump(doc, open(f"{out}/skinned_pmc_summary.json","w"), indent=1)
print(json.dumps(doc, indent=1)[:3000])
PY
for v in palette bounds; do echo "== $v"; head -6 $OUT/kernel_stats_$v.csv | cut -c1-160; done
