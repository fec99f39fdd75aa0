#!/bin/bash
# Round 4: the second translation unit (per-triangle stage, views, skinning, light lists) built with the max-ILP scheduling
# strategy (as the api units are) against the product build. usage: tools/r04_stages_ilp_ab.sh > gpurun_out/r04_stages_ilp_ab.txt
V=renderer_amd/lib/libmip_w5_triilp.so
for rep in 1 2; do
  for lib in "" $V; do
    echo "== lib=${lib:-product} rep=$rep"
    MIP_LIBRARY=$lib python3 tools/tri_bench.py 2 100000 2>&1 | tail -1
    MIP_LIBRARY=$lib python3 tools/tri_bench.py 2 100000 strips 2>&1 | tail -1
    MIP_LIBRARY=$lib python3 tools/skin_bench.py 256000 50 2>&1 | tail -2
    MIP_LIBRARY=$lib python3 tools/light_bench.py 2>&1 | tail -3
  done
done
python3 tools/views_kbench.py 1000000 --libs default,$V 2>&1 | tail -12
