// instance_pipeline_kernels.hpp — every gfx950 kernel of the library, one header per subsystem.
// This TU must be compiled with -ffp-contract=off and without fast-math (see instance_kernel.hpp).
#pragma once

#include "instance_kernel.hpp"       // rows a-1 .. a-7: matrices, world AABB, frustum test, commands + compaction (+ TLAS rows)
#include "merge_kernel.hpp"          // row e: merge of the all-gathered shard draw lists
#include "triangle_kernels.hpp"      // row f-1: per-triangle cull + index-stream append, re-compaction
#include "light_lists_kernel.hpp"    // row f-4: per-light shadow-pass draw lists
#include "skinning_kernel.hpp"       // extension (BASELINE config 5): joint palette + posed box
#include "views_kernel.hpp"          // row f-4: up to four culled views (per-light lists, cascades) in one launch
