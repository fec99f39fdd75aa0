// instance_pipeline_kernels.hpp — the gfx950 kernels mip_api.hip compiles itself, one header per subsystem
// (the per-triangle stage is a translation unit of its own: triangle_tu.hip).
// This TU must be compiled with -ffp-contract=off and without fast-math (see instance_kernel.hpp).
#pragma once

#include "instance_kernel.hpp"       // rows a-1 .. a-7: matrices, world AABB, frustum test, commands + compaction (+ TLAS rows)
#include "merge_kernel.hpp"          // row e: merge of the all-gathered shard draw lists
#include "triangle_args.hpp"         // row f-1: argument blocks + launchers; the kernels live in triangle_tu.hip (own flags)
#include "light_lists_kernel.hpp"    // row f-4: per-light shadow-pass draw lists
#include "skinning_kernel.hpp"       // extension (BASELINE config 5): joint palette + posed box
#include "views_kernel.hpp"          // row f-4: up to four culled views (per-light lists, cascades) in one launch
