#!/bin/bash
# Round-5 profile set (run on the GPU box in two halves: tools/r05_profiles.sh a | b; a gpurun call is limited to 20 minutes).
# Every step writes under gpurun_out/; tools/r05_collect.sh copies what is to be judged into profiles/.
#  a1  rocprofv3 kernel stats of the default bench command (config 3) and of config 2
#  a2  FETCH/WRITE PMC passes of the frame kernel at 1 M (config 3) AND at 10 M (config 4: the honest HBM figure), SQ counters at 1 M / 100 k
#  a3  both frame-kernel orders either side of the crossover (what a launch of mixed orders could gain at 1 M)
#  b1  PMC summaries of the other kernels whose bench legs read them: per-triangle stage (mip_triangle_stage_kernel; one-mesh 100 k rows / strips: the size-sorted
#      wave-per-command kernel; mixed 100 k: the range kernel), four views, skinned frame
#  b2  the N > 1 bench path rehearsed with 2 and 3 ranks on one GPU over gloo (cpu_baseline on the line), the default bench line
set -o pipefail
mkdir -p gpurun_out/r05
step() { echo "== $1 ($(date +%T))"; }
half=${1:-a}
if [ "$half" = a ]; then
step "a1 kernel stats, configs 3 and 2"
bash tools/profile.sh r05_cfg3 > gpurun_out/r05/profile_cfg3.log 2>&1; tail -4 gpurun_out/r05/profile_cfg3.log
BENCH_ARGS="--config 2" bash tools/profile.sh r05_cfg2 > gpurun_out/r05/profile_cfg2.log 2>&1; tail -3 gpurun_out/r05/profile_cfg2.log
step "a2 PMC traffic 1 M and 10 M, SQ counters"
bash tools/pmc.sh r05_cfg3 3 > gpurun_out/r05/pmc_cfg3.log 2>&1; tail -4 gpurun_out/r05/pmc_cfg3.log
bash tools/pmc.sh r05_cfg4 4 > gpurun_out/r05/pmc_cfg4.log 2>&1; tail -4 gpurun_out/r05/pmc_cfg4.log
bash tools/pmc_sq.sh r05_cfg3 3 > gpurun_out/r05/pmc_sq_cfg3.log 2>&1; tail -2 gpurun_out/r05/pmc_sq_cfg3.log
bash tools/pmc_sq.sh r05_cfg2 2 > gpurun_out/r05/pmc_sq_cfg2.log 2>&1; tail -2 gpurun_out/r05/pmc_sq_cfg2.log
step "a3 both orders around the crossover"
for order in 1 3; do
  echo "--- MIP_TUNE_ORDER=$order"
  MIP_TUNE_ORDER=$order timeout -k 10 300 python tools/kbench.py --configs 3,3,3,3 --n 524288,786432,1000000,1500000 --libs default 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05/orders_around_1m.txt; tail -4 gpurun_out/r05/orders_around_1m.txt
else
step "b1 PMC of the other kernels"
PMC_TRI_KERNEL="mip_triangle_stage_kernel(" bash tools/pmc_tri.sh r05_tri_rows 2 100000 rows > gpurun_out/r05/pmc_tri_rows.log 2>&1; tail -2 gpurun_out/r05/pmc_tri_rows.log
PMC_TRI_KERNEL="mip_triangle_stage_kernel(" bash tools/pmc_tri.sh r05_tri_strips 2 100000 strips > gpurun_out/r05/pmc_tri_strips.log 2>&1; tail -2 gpurun_out/r05/pmc_tri_strips.log
PMC_TRI_KERNEL="mip_triangle_stage_kernel(" bash tools/pmc_tri.sh r05_tri_mixed 3 100000 rows > gpurun_out/r05/pmc_tri_mixed.log 2>&1; tail -2 gpurun_out/r05/pmc_tri_mixed.log
bash tools/pmc_views.sh r05_views 1000000 > gpurun_out/r05/pmc_views.log 2>&1; tail -3 gpurun_out/r05/pmc_views.log
bash tools/pmc_skin.sh r05_skin > gpurun_out/r05/pmc_skin.log 2>&1; tail -6 gpurun_out/r05/pmc_skin.log
step "b2 rehearsal, default bench"
for R in 2 3; do
  MIP_BENCH_BACKEND=gloo MIP_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $R \
    --master-addr 127.0.0.1 --master-port $((29633 + R)) bench.py --gpus $R --steps 5 --warmup 2 --no-extra --cpu-seconds 5 \
    > gpurun_out/r05_bench_rehearsal_${R}ranks_one_gpu_gloo.json 2> gpurun_out/r05/bench_rehearsal_${R}ranks.err
  echo "ranks $R rc=$?"
  python3 -c "
import json
d=json.load(open('gpurun_out/r05_bench_rehearsal_${R}ranks_one_gpu_gloo.json'))
print(d['n_gpus'], d['ms_per_step'], d['config'].get('prefix_helps_max_over_ranks'), d['config']['commands_total'], (d.get('cpu_baseline') or {}).get('value'))
" || tail -5 gpurun_out/r05/bench_rehearsal_${R}ranks.err
done
timeout -k 10 400 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.load(open('gpurun_out/r05/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
for k,v in d['extra'].items():
    if isinstance(v, dict):
        print(k, {kk:(round(vv,5) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in('ms_per_step','ms_per_frame','ms_per_launch','frac_of_8000','error')}, (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('traffic'))
PY
fi
step done
