set -o pipefail
mkdir -p gpurun_out/r03
bash tools/profile.sh r03_cfg3 > gpurun_out/r03/profile_cfg3.log 2>&1; tail -3 gpurun_out/r03/profile_cfg3.log
python3 -c "import json; d=json.load(open('gpurun_out/prof_r03_cfg3/bench_under_rocprof.json')); print('under rocprof:', d['ms_per_step'], d['roofline']['frac'])"
bash tools/pmc.sh r03_cfg3 3 > gpurun_out/r03/pmc_cfg3.log 2>&1; tail -5 gpurun_out/r03/pmc_cfg3.log
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; echo "bench rc=$?"
python3 -c "
import json
d=json.load(open('gpurun_out/r03/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
for k,v in d['extra'].items():
    print(k, {kk:(round(vv,5) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in('ms_per_step','ms_per_frame','ms_per_launch','frac_of_8000','error')}, (v.get('roofline') or {}).get('frac'))
"
