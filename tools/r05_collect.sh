#!/bin/bash
# copies what tools/r05_profiles.sh left under gpurun_out/ into profiles/ (tracked) under the names DESIGN.md and bench.py cite
set -e
G=gpurun_out; P=profiles
for c in 2 3; do
  cp $G/prof_r05_cfg$c/kernel_stats.csv $P/r05_cfg${c}_serialized_kernel_stats.csv
  cp $G/prof_r05_cfg$c/bench_under_rocprof.json $P/r05_cfg${c}_serialized_bench_under_rocprof.json
  cp $G/pmc_sq_r05_cfg$c/sq_summary.json $P/r05_cfg${c}_sq_counters.json
done
cp $G/pmc_r05_cfg3/pmc_summary.json $P/r05_cfg3_pmc_summary.json
cp $G/pmc_r05_cfg4/pmc_summary.json $P/r05_cfg4_10m_pmc_summary.json
cp $G/r05/orders_around_1m.txt $P/r05_orders_around_1m.txt
cp $G/pmc_r05_tri_rows/tri_pmc_summary.json $P/r05_triangle_cull_100k_rows_pmc_summary.json
cp $G/pmc_r05_tri_strips/tri_pmc_summary.json $P/r05_triangle_cull_100k_strips_pmc_summary.json
cp $G/pmc_r05_tri_mixed/tri_pmc_summary.json $P/r05_triangle_cull_mixed_100k_pmc_summary.json
cp $G/pmc_r05_views/views_pmc_summary.json $P/r05_views_x4_pmc_summary.json
cp $G/pmc_r05_skin/skinned_pmc_summary.json $P/r05_skinned_pmc_summary.json
for r in 2 3; do cp $G/r05_bench_rehearsal_${r}ranks_one_gpu_gloo.json $P/r05_bench_rehearsal_${r}ranks_one_gpu_gloo.json; done
cp $G/r05/bench_default.json $P/r05_bench_default.json
ls -la $P | grep r05_ | wc -l
