#!/bin/bash
# PMC passes (one counter group per run, no tracing domains besides kernel-trace).
# usage: tools/pmc.sh <tag> <config>
TAG=${1:-r01}; CFG=${2:-3}
OUT=$(pwd)/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
for v in full no_cmds bitmap_only model_only; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${v}_$c -- python3 $ROOT/tools/pmc_run.py $v $CFG 10 > $OUT/${v}_$c.log 2>&1
  done
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
out="$OUT"
import json, sys
sys.path.insert(0, "$ROOT")
from renderer_amd import scene
cfg=$CFG
rows=[]
for v in ("full","no_cmds","bitmap_only","model_only"):
    row={}
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        fs=glob.glob(f"{out}/{v}_{c}/**/*counter_collection.csv", recursive=True)
        vals=[]
        for f in fs:
            for r in csv.DictReader(open(f)):
                if "mip_instance" in r["Kernel_Name"] and r["Counter_Name"]==c:
                    vals.append(float(r["Counter_Value"]))
        row[c]=(sum(vals)/len(vals) if vals else float("nan"), len(vals))
    print(v, {k:(round(a,1),n) for k,(a,n) in row.items()})
    f,w=row["FETCH_SIZE"][0],row["WRITE_SIZE"][0]
    rows.append(dict(config=cfg, instances=scene.CONFIGS[cfg]["n"], variant=v, launches=row["FETCH_SIZE"][1],
                     FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w,
                     fetch_bytes_corrected=2*f*1024, write_bytes=w*1024, hbm_bytes_per_launch=2*f*1024+w*1024))
import hashlib
sha=hashlib.sha256()
for f in ("instance_kernel.hpp",): sha.update(open("$ROOT/renderer_amd/csrc/"+f,"rb").read())
json.dump(dict(kernel_source_sha=sha.hexdigest()[:16], note="rocprofv3 --pmc, one counter per pass; FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read stream, MI355X_MICROARCH.md HBM section; calibrated here: bitmap_only reads 36 B/instance), WRITE_SIZE exact", workloads=rows), open(f"{out}/pmc_summary.json","w"), indent=1)
PY
