#!/bin/bash
# round 4: the bench's N > 1 path rehearsed with 2 and 3 ranks ON ONE GPU over gloo (RCCL refuses two ranks on one device): two or three
# spin-waiting shard kernels of different PROCESSES share the chip — the situation that deadlocked round 2's kernel and that round 3
# survived through a 0.5 s time-out. Numbers are not measurements; what matters: rc 0, no error, and prefix_helps in the line.
mkdir -p gpurun_out
for R in 2 3; do
  MIP_BENCH_BACKEND=gloo MIP_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $R \
    --master-addr 127.0.0.1 --master-port $((29533 + R)) bench.py --gpus $R --steps 5 --warmup 2 --no-extra --no-cpu-baseline \
    > gpurun_out/r04_bench_rehearsal_${R}ranks_one_gpu_gloo.json 2> gpurun_out/r04_bench_rehearsal_${R}ranks.err
  echo "ranks $R rc=$?"
  python3 -c "
import json
d=json.load(open('gpurun_out/r04_bench_rehearsal_${R}ranks_one_gpu_gloo.json'))
print(d['n_gpus'], d['ms_per_step'], d['config'].get('prefix_helps_max_over_ranks'), d['config']['commands_total'], d.get('breakdown_ms_per_step'))
" || tail -5 gpurun_out/r04_bench_rehearsal_${R}ranks.err
done
