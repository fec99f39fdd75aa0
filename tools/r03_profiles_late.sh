# late round 3 (after the second translation unit took over views / light lists / skinning / the commands-first frame kernel):
# everything of tools/r03_profiles.sh except the per-triangle passes, whose kernel source did not change
set -o pipefail
mkdir -p gpurun_out/r03
bash tools/profile.sh r03_cfg3 > gpurun_out/r03/profile_cfg3.log 2>&1; tail -6 gpurun_out/r03/profile_cfg3.log
python3 -c "import json; d=json.load(open('gpurun_out/prof_r03_cfg3/bench_under_rocprof.json')); print('under rocprof:', d['ms_per_step'], d['roofline']['frac'])"
BENCH_ARGS="--config 2" bash tools/profile.sh r03_cfg2 > gpurun_out/r03/profile_cfg2.log 2>&1; tail -4 gpurun_out/r03/profile_cfg2.log
bash tools/pmc.sh r03_cfg3 3 > gpurun_out/r03/pmc_cfg3.log 2>&1; tail -5 gpurun_out/r03/pmc_cfg3.log
bash tools/pmc_sq.sh r03_cfg3 3 > gpurun_out/r03/pmc_sq_cfg3.log 2>&1; tail -3 gpurun_out/r03/pmc_sq_cfg3.log
bash tools/pmc_sq.sh r03_cfg2 2 > gpurun_out/r03/pmc_sq_cfg2.log 2>&1; tail -3 gpurun_out/r03/pmc_sq_cfg2.log
bash tools/pmc_skin.sh r03_skin > gpurun_out/r03/pmc_skin.log 2>&1; tail -12 gpurun_out/r03/pmc_skin.log
bash tools/pmc_views.sh r03_views 1000000 > gpurun_out/r03/pmc_views.log 2>&1; tail -12 gpurun_out/r03/pmc_views.log
