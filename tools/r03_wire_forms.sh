mkdir -p gpurun_out/r03
for w in 2 1 0; do
  MIP_BENCH_WIRE=$w MIP_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 10 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r03/bench_dist1_form$w.json 2> gpurun_out/r03/bench_dist1_form$w.err; echo "form $w rc=$?"
done
python - <<'PY'
import json
for f in (2,1,0):
    d = json.load(open(f"gpurun_out/r03/bench_dist1_form{f}.json"))
    print(f, round(d["ms_per_step"],5), d["config"]["chunk_bytes_per_rank"], d["config"]["chunk_format"][:30], {k:round(v,5) for k,v in d["breakdown_ms_per_step"].items() if k!="note"})
PY
