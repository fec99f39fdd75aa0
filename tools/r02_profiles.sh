# round-2 profile set for the headline (BASELINE config 3): rocprofv3 kernel stats of the default bench command,
# FETCH/WRITE PMC passes, SQ counters
set -o pipefail
mkdir -p gpurun_out/r02
bash tools/profile.sh r02_cfg3 > gpurun_out/r02/profile_cfg3.log 2>&1; tail -12 gpurun_out/r02/profile_cfg3.log
cat gpurun_out/prof_r02_cfg3/bench_under_rocprof.json | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['roofline']['frac'])"
bash tools/pmc.sh r02_cfg3 3 > gpurun_out/r02/pmc_cfg3.log 2>&1; tail -6 gpurun_out/r02/pmc_cfg3.log
bash tools/pmc_sq.sh r02_cfg3 3 > gpurun_out/r02/pmc_sq_cfg3.log 2>&1; tail -40 gpurun_out/r02/pmc_sq_cfg3.log
