#!/bin/bash
# per-kernel times of a large ordered-tiles frame (three launches), 1 M and 10 M: rocprofv3 kernel trace of tools/kbench.py
export TMPDIR=/tmp; ROOT=$(pwd); rm -rf gpurun_out/oprof; mkdir -p gpurun_out/oprof; cd /tmp
MIP_TUNE_ORDERED_TILES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/oprof -- python3 $ROOT/tools/kbench.py --child --configs ${1:-3,4} --n ${2:-1000000,10000000} --subset full > $ROOT/gpurun_out/oprof/log.txt 2>&1
cd $ROOT
cat > /tmp/oprof.py <<'PY'
import csv,sys,collections,glob
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"][:72], r["Grid_Size_X"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items()):
    if len(v)>50:
        v.sort(); print(k, len(v), "avg %.2f us  median %.2f  p10 %.2f  p90 %.2f" % (sum(v)/len(v)/1000, v[len(v)//2]/1000, v[len(v)//10]/1000, v[len(v)*9//10]/1000))
PY
python3 /tmp/oprof.py gpurun_out/oprof
grep -v "amdgpu\|rocprofv3\|^W\|^E" gpurun_out/oprof/log.txt | tail -3
