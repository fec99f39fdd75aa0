// instance_pipeline_kernels.hpp — what mip_api.hip compiles itself: the stores-first frame kernel and its helpers, the
// shard merge, and the argument blocks + launchers of everything built in the second translation unit (stages_tu.hip).
// Both units must be compiled with -ffp-contract=off and without fast-math (see instance_kernel.hpp).
#pragma once

#include "instance_kernel.hpp"       // rows a-1 .. a-7: matrices, world AABB, frustum test, commands + compaction (+ TLAS rows)
#include "merge_kernel.hpp"          // row e: merge of the all-gathered shard draw lists
#include "stage_args.hpp"            // rows f-1, f-4, skinning, commands-first frame kernel: built in stages_tu.hip (own flags)
