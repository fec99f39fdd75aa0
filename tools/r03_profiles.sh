# round-3 profile set (run on the GPU box): rocprofv3 kernel stats of the default bench command, FETCH/WRITE PMC passes and SQ
# counters for the headline (config 3) and for BASELINE configs[1] (config 2, 100 k), the triangle kernel on both mesh layouts,
# the skinned extension. Summaries land in gpurun_out/; copy what is cited into profiles/.
set -o pipefail
mkdir -p gpurun_out/r03
bash tools/profile.sh r03_cfg3 > gpurun_out/r03/profile_cfg3.log 2>&1; tail -6 gpurun_out/r03/profile_cfg3.log
python3 -c "import json; d=json.load(open('gpurun_out/prof_r03_cfg3/bench_under_rocprof.json')); print('under rocprof:', d['ms_per_step'], d['roofline']['frac'])"
BENCH_ARGS="--config 2" bash tools/profile.sh r03_cfg2 > gpurun_out/r03/profile_cfg2.log 2>&1; tail -4 gpurun_out/r03/profile_cfg2.log
bash tools/pmc.sh r03_cfg3 3 > gpurun_out/r03/pmc_cfg3.log 2>&1; tail -5 gpurun_out/r03/pmc_cfg3.log
bash tools/pmc_sq.sh r03_cfg3 3 > gpurun_out/r03/pmc_sq_cfg3.log 2>&1; tail -3 gpurun_out/r03/pmc_sq_cfg3.log
bash tools/pmc_sq.sh r03_cfg2 2 > gpurun_out/r03/pmc_sq_cfg2.log 2>&1; tail -3 gpurun_out/r03/pmc_sq_cfg2.log
bash tools/pmc_tri.sh r03_tri_rows 2 100000 rows > gpurun_out/r03/pmc_tri_rows.log 2>&1; tail -3 gpurun_out/r03/pmc_tri_rows.log
bash tools/pmc_tri.sh r03_tri_strips 2 100000 strips > gpurun_out/r03/pmc_tri_strips.log 2>&1; tail -3 gpurun_out/r03/pmc_tri_strips.log
bash tools/pmc_skin.sh r03_skin > gpurun_out/r03/pmc_skin.log 2>&1; tail -12 gpurun_out/r03/pmc_skin.log
