#!/bin/bash
# round-4 profile set (run on the GPU box; ~15 min): the artefacts profiles/r04_* are copied from. Every step writes under gpurun_out/.
#  1  order independence: tools/r04_selfhelp_check.py (diagnostic build, permuted tile order, fault injection)
#  2  same-box A/B against round 3's library (renderer_amd/lib/libmip_r03.so, built from the round-3 commit)
#  3  rocprofv3 kernel stats of the default bench command (config 3) and of config 2; FETCH/WRITE PMC passes and SQ counters
#  4  PMC summaries of the kernels whose bench legs read them: triangle stage (both layouts), four views, skinned frame
#  5  the wire merge (timing + kernel trace), the semaphore hand-over, the VALU issue ceiling, the multi-process rehearsal
#  6  the default bench line
set -o pipefail
mkdir -p gpurun_out/r04
step() { echo "== $1 ($(date +%T))"; }
step "1 order independence"
timeout -k 10 700 python tools/r04_selfhelp_check.py > gpurun_out/r04/selfhelp_any_order.txt 2>&1; echo "rc=$?"; tail -2 gpurun_out/r04/selfhelp_any_order.txt
step "2 A/B vs round 3"
if [ -f renderer_amd/lib/libmip_r03.so ]; then bash tools/r04_ab.sh > gpurun_out/r04/ab_vs_r03.log 2>&1; cp gpurun_out/r04_kbench_ab.txt gpurun_out/r04/selfhelp_ab_vs_r03.txt; tail -3 gpurun_out/r04/selfhelp_ab_vs_r03.txt; fi
step "3 kernel stats + PMC, configs 3 and 2"
bash tools/profile.sh r04_cfg3 > gpurun_out/r04/profile_cfg3.log 2>&1; tail -4 gpurun_out/r04/profile_cfg3.log
BENCH_ARGS="--config 2" bash tools/profile.sh r04_cfg2 > gpurun_out/r04/profile_cfg2.log 2>&1; tail -3 gpurun_out/r04/profile_cfg2.log
bash tools/pmc.sh r04_cfg3 3 > gpurun_out/r04/pmc_cfg3.log 2>&1; tail -4 gpurun_out/r04/pmc_cfg3.log
bash tools/pmc_sq.sh r04_cfg3 3 > gpurun_out/r04/pmc_sq_cfg3.log 2>&1; tail -2 gpurun_out/r04/pmc_sq_cfg3.log
bash tools/pmc_sq.sh r04_cfg2 2 > gpurun_out/r04/pmc_sq_cfg2.log 2>&1; tail -2 gpurun_out/r04/pmc_sq_cfg2.log
step "4 PMC of the other kernels"
bash tools/pmc_tri.sh r04_tri_rows 2 100000 rows > gpurun_out/r04/pmc_tri_rows.log 2>&1; tail -2 gpurun_out/r04/pmc_tri_rows.log
bash tools/pmc_tri.sh r04_tri_strips 2 100000 strips > gpurun_out/r04/pmc_tri_strips.log 2>&1; tail -2 gpurun_out/r04/pmc_tri_strips.log
bash tools/pmc_views.sh r04_views 1000000 > gpurun_out/r04/pmc_views.log 2>&1; tail -3 gpurun_out/r04/pmc_views.log
bash tools/pmc_skin.sh r04_skin > gpurun_out/r04/pmc_skin.log 2>&1; tail -14 gpurun_out/r04/pmc_skin.log
step "5 merge, semaphores, VALU ceiling, rehearsal"
bash tools/r04_merge.sh > gpurun_out/r04/merge.log 2>&1; tail -8 gpurun_out/r04/merge.log
for n in 1000000 100000; do timeout -k 10 100 renderer_amd/lib/mip_semaphore_bench $n 2000; done > gpurun_out/r04/semaphore_handover.jsonl 2>&1; cat gpurun_out/r04/semaphore_handover.jsonl
[ -x tools/micro/valu_issue ] && timeout -k 10 200 tools/micro/valu_issue > gpurun_out/r04/valu_issue.txt 2>&1; tail -2 gpurun_out/r04/valu_issue.txt
bash tools/r04_rehearsal.sh > gpurun_out/r04/rehearsal.log 2>&1; cat gpurun_out/r04/rehearsal.log
step "6 default bench"
timeout -k 10 290 python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.load(open('gpurun_out/r04/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
for k,v in d['extra'].items():
    if isinstance(v, dict):
        print(k, {kk:(round(vv,5) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in('ms_per_step','ms_per_frame','ms_per_launch','frac_of_8000','error')}, (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('traffic'))
PY
step done
