# round-3 bench lines: the default line, the N>1 path rehearsed with one rank over real RCCL (wire form and 20-byte form),
# and the N>1 control flow with two ranks on ONE GPU over gloo (bench.py launches its own ranks: no torchrun here)
set -o pipefail
mkdir -p gpurun_out/r03
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r03/bench_default.json
MIP_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 10 --warmup 5 > gpurun_out/r03/bench_dist1_wire.json 2> gpurun_out/r03/bench_dist1_wire.err; echo "dist wire rc=$?"
MIP_BENCH_WIRE=0 MIP_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 10 --warmup 5 --no-extra > gpurun_out/r03/bench_dist1_cmds.json 2> gpurun_out/r03/bench_dist1_cmds.err; echo "dist cmds rc=$?"
python - <<'PY'
import json
for f in ("wire", "cmds"):
    try:
        d = json.load(open(f"gpurun_out/r03/bench_dist1_{f}.json"))
        print(f, d["ms_per_step"], d["config"]["chunk_bytes_per_rank"], d["config"].get("chunk_format"), d.get("breakdown_ms_per_step"))
    except Exception as e:
        print(f, "failed:", e)
PY
# two ranks on one GPU: `python bench.py --gpus 2` must refuse on a 1-GPU box (no line, non-zero) ...
python bench.py --gpus 2 --steps 2 > gpurun_out/r03/bench_gpus2_refused.out 2> gpurun_out/r03/bench_gpus2_refused.err; echo "gpus2 on a 1-GPU box: rc=$? stdout bytes=$(stat -c %s gpurun_out/r03/bench_gpus2_refused.out)"; tail -2 gpurun_out/r03/bench_gpus2_refused.err
# ... and the rehearsal with both ranks on device 0 over gloo goes through torchrun explicitly
MIP_BENCH_DEVICE=0 MIP_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 --no-extra > gpurun_out/r03/bench_rehearsal_2ranks_gloo.json 2> gpurun_out/r03/bench_rehearsal_2ranks_gloo.err; echo "2-rank gloo rehearsal rc=$?"; tail -c 1200 gpurun_out/r03/bench_rehearsal_2ranks_gloo.json
