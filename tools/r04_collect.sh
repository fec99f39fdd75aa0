#!/bin/bash
# copies what tools/r04_profiles.sh left under gpurun_out/ into profiles/ (tracked) under the names DESIGN.md and bench.py cite
set -e
G=gpurun_out; P=profiles
cp $G/r04/selfhelp_any_order.txt $P/r04_selfhelp_any_order.txt
cp $G/r04/selfhelp_ab_vs_r03.txt $P/r04_selfhelp_ab_vs_r03.txt
for c in 2 3; do
  cp $G/prof_r04_cfg$c/kernel_stats.csv $P/r04_cfg${c}_serialized_kernel_stats.csv
  cp $G/prof_r04_cfg$c/bench_under_rocprof.json $P/r04_cfg${c}_serialized_bench_under_rocprof.json
  cp $G/pmc_sq_r04_cfg$c/sq_summary.json $P/r04_cfg${c}_sq_counters.json
done
cp $G/pmc_r04_cfg3/pmc_summary.json $P/r04_cfg3_pmc_summary.json
cp $G/pmc_r04_tri_rows/tri_pmc_summary.json $P/r04_triangle_cull_100k_rows_pmc_summary.json
cp $G/pmc_r04_tri_strips/tri_pmc_summary.json $P/r04_triangle_cull_100k_strips_pmc_summary.json
cp $G/pmc_r04_views/views_pmc_summary.json $P/r04_views_x4_pmc_summary.json
cp $G/pmc_r04_skin/skinned_pmc_summary.json $P/r04_skinned_pmc_summary.json
cp $G/pmc_r04_skin/kernel_stats_palette.csv $P/r04_skinned_kernel_stats_with_palette.csv
cp $G/pmc_r04_skin/kernel_stats_bounds.csv $P/r04_skinned_kernel_stats_bounds_only.csv
for r in 2 3; do cp $G/r04_bench_rehearsal_${r}ranks_one_gpu_gloo.json $P/r04_bench_rehearsal_${r}ranks_one_gpu_gloo.json; done
cp $G/r04/bench_default.json $P/r04_bench_default.json
sed -i 's/\r//' $P/r04_*.txt
ls -la $P | grep r04_ | wc -l
