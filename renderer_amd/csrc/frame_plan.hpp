// frame_plan.hpp — WHICH kernels a frame runs, in which instantiation, over which grid, and what scratch they need: a
// PURE function of the context's state and of what the caller asked for. No HIP, no allocation, no I/O — so the
// decision table of the library can be enumerated on a CPU, under the sanitizers, without a GPU
// (tests/test_frame_plan.py builds tests/frame_plan_check.cpp with gcc -fsanitize=address,undefined). api_frame.hip
// executes the plan it gets; it takes no decision of its own.
//
// Replaces the mode selection that rounds 1-3 had interleaved with the launches inside run_frame (mip_api.hip).
#pragma once

#include <stdint.h>

#include "../../include/mi_instance_pipeline.h"

namespace mip {

#ifndef MIP_TILE
#define MIP_TILE 256
#endif
constexpr uint32_t kPlanTile = MIP_TILE;          // = kTile (instance_kernel.hpp; static_assert in api_frame.hip)
constexpr uint32_t kPlanTriParts = 16;            // = kTriParts
constexpr uint32_t kPlanTriPartMaxT = 8;          // = kTriPartMaxT

// What the decision depends on: resident state + tuning knobs of the context.
struct PlanState {
  uint32_t n = 0;                  // resident instances
  uint32_t n_meshes = 0;
  uint32_t max_instances = 0;
  uint32_t cu_count = 256;
  uint32_t frame_slots = 1;        // MipConfig.frames_in_flight
  bool have_instances = false, have_meshes = false, have_geometry = false;
  bool nonfinite = false;          // the upload-time census found an instance that needs the fall-back arithmetic tiers
  bool force_general = false;      // MIP_TUNE_FORCE_GENERAL
  int force_order = 0;             // MIP_TUNE_ORDER: 1 | 3, 0 = by launch size
  uint32_t tri_block_threads = 0;  // MIP_TUNE_TRI_BLOCK_THREADS: 256 | 512 | 1024, 0 = by instance count
  uint32_t tri_block_max = 65536;  // instance counts up to this use the workgroup-per-command triangle kernel
  uint32_t tri_parts_max = 1024;   // instance counts up to this use the parts kernel, 0 = off
  uint32_t tri_chunks_from = 0;    // instance counts from this use the chunk kernel (round 5); 0xffffffff = never (MIP_TUNE_TRI_CHUNKS_FROM)
  uint32_t tri_chunk_blocks_per_cu = 8;  // workgroups of the chunk kernel a CU holds at once (asked of the runtime)
  bool tri_no_choice = false;      // MIP_TUNE_TRI_NO_CHOICE: large frames always take the wave-per-command kernel (A/B)
  uint32_t max_lod_tris = 0;       // largest triangle count of LOD 0 / LOD 1 over the mesh table
  uint32_t n_joints = 0;           // skinned frames
};

// What the caller asked for (MipOutputs, decoded) + how the frame was entered.
struct PlanRequest {
  bool model = false, bitmap = false, cmds = false, count = false, index_total = false, aabb = false, tlas = false;
  bool triangles = false;          // culled_index_buffer set
  bool skinned = false;            // mip_run_skinned
  uint32_t flags = 0;              // MIP_OUT_*
  uintptr_t cmds_address = 0;      // draw_cmds as an integer (alignment rules of the wire forms)
};

enum class TriangleKernel : uint8_t { none, parts, block, waves, chunks, sorted };
enum class Recompact : uint8_t { none, single, wide };

struct LaunchPlan {
  int32_t status = MIP_OK;         // MIP_OK, or the error the frame must be refused with ...
  const char* why = "";            // ... and its message (static text; api_frame.hip adds the numbers)
  bool device_out = false, async = false;
  bool empty = false;              // no resident instances: counts are zeroed, nothing launches
  // frame kernel (rows a-1 .. a-7)
  uint32_t n_tiles = 0;            // grid of the frame kernel
  int order = 1;                   // kOrder: 1 = stores first (mip_api/api_frame unit), 3 = commands first (stages_tu unit)
  bool general = false;            // kGeneral: the kernel that carries the fall-back arithmetic tiers
  bool box_override = false;       // kBoxOverride: skinned frame
  int wire = 0;                    // kWire: 0 | 1 | 2 = packed
  uint32_t group_shift = 4;        // log2(tiles per level-1 group)
  bool uses_prefix_state = false;  // the launch emits commands: it needs a fresh tag
  // extension: skinning kernel in front of the frame kernel
  bool skin = false;
  uint32_t skin_blocks = 0;
  // row f-1: per-triangle stage behind it
  TriangleKernel tri = TriangleKernel::none;
  uint32_t tri_threads = 0, tri_blocks = 0;
  uint32_t tri_map_blocks = 0;     // range kernel: grid of the range-map kernel in front of it (one thread per command); the sort kernels' grid too
  bool tri_block_tickets = false;  // the workgroup-per-command kernel pulls its commands from the counter (many more commands than workgroups)
  uint32_t tri_either_blocks = 0;  // > 0: ALSO launch the 256-thread workgroup-per-command kernel over this grid; the two kernels pick
                                   // one of themselves on the device from the frame's own totals (triangle_kernels.hpp, tri_choice)
  bool tri_reset_ticket = false;   // the wave-per-command kernel pulls commands from a counter the host zeroes
  Recompact recompact = Recompact::none;
  uint32_t recompact_blocks = 0;
  // scratch the slot must own before the launches
  bool need_staging = false;       // host outputs: device staging + copy-back
  bool need_tri_scratch = false;   // list before re-compaction, source offsets, block counts
  bool need_part_status = false;   // granules of the parts kernel
  bool need_chunk_scratch = false; // chunk map + granules of the chunk kernel
  bool need_skin_box = false;      // per-instance posed box
};

inline uint32_t plan_tiles_for(uint32_t n) { return (n + kPlanTile - 1u) / kPlanTile; }

inline int plan_wire_form(uint32_t out_flags) { return (out_flags & MIP_OUT_WIRE) ? ((out_flags & MIP_OUT_WIRE_PACKED) ? 2 : 1) : 0; }

// mip_wire_index_bits, restated here so that the plan has no link-time dependency (test_abi.py checks both against the header's formula)
inline uint32_t plan_wire_index_bits(uint32_t n_meshes) {
  uint32_t mesh_bits = 0;
  while (mesh_bits < 31u && (1ull << mesh_bits) < n_meshes) ++mesh_bits;
  return 31u - mesh_bits;
}

inline LaunchPlan plan_refuse(int32_t status, const char* why) {
  LaunchPlan p;
  p.status = status;
  p.why = why;
  return p;
}

// Large per-triangle frames launch the wave-per-command AND the workgroup-per-command grid (LaunchPlan.tri_either_blocks); every
// workgroup of both evaluates THIS on the device, from totals the frame kernel has just written, and one grid returns at once.
// (constexpr: callable from device code as it stands, and enumerated on the CPU by tests/native/frame_plan_check.cpp.)
// A wave walks its command alone: the launch cannot end before the largest command has been walked by ONE wave, so when that
// walk is long against a wave's share of the whole frame (total triangles / the 8 192 waves of the grid) the launch is mostly
// tail, and a workgroup per command — a quarter of the walk, no tickets, but a barrier per step — wins. Measured
// (profiles/r04_triangle_kernel_choice.txt), share = largest command / (total / 8 192), spread = largest / mean command:
//   (workgroup grid pulling tickets; ms per frame, workgroup-per-command / wave-per-command)
//   mixed scene (spread ~10): 100 k instances share 3.3: 0.43 / 0.71; 200 k 1.6: 0.75 / 1.03; 400 k 0.8: 1.38 / 1.73;
//                             1 M 0.33: 3.31 / 3.70
//   one-mesh scene (spread 1.9: every command is near the largest, the tail is only rounding): 70 k share 0.85: 0.755 / 0.776;
//                             100 k 0.6: 1.03 / 1.01; 150 k 0.4: 1.42 / 1.40; 300 k 0.2: 2.63 / 2.51
// Rule: share > 0.7, or spread > 4.
// Round 5, frames above tri_block_max instances: the wave-per-command kernel takes its commands by size class, largest first, and
// asks for issue priority while it has walked few of them (triangle_kernels.hpp), so what is left against it is a frame whose largest
// command is long against a wave's share of the whole frame (share = largest command x waves / triangles of the frame) AND far above
// the mean command (spread = largest x commands / triangles): then the range kernel, which cuts commands, takes it. Measured
// (profiles/r05_triangle_stage_modes.txt, final build; ms per frame, range decomposition / wave-per-command over sorted commands):
//   mixed scene (spread ~10): 70 k instances share 4.7: 0.278 / 0.359; 100 k 3.3: 0.358 / 0.422; 130 k 2.55: 0.441 / 0.452;
//                             160 k 2.07: 0.528 / 0.505; 200 k 1.6: 0.653 / 0.591; 400 k 0.8: 1.164 / 1.040; 600 k 0.55: 1.66 / 1.47
//   one-mesh scene (spread 1.9): 70 k share 0.85: 0.788 / 0.689; 100 k 0.6: 1.052 / 0.925; 300 k 0.2: 2.77 / 2.38
// Rule: share > 2.25 and spread > 3. Evaluated on the device by every workgroup of the stage's grid (from the slot's own command list).
constexpr bool plan_tri_choice_is_ranges(uint32_t max_lod_tris, uint32_t total_tris, uint32_t command_count, uint32_t n_waves) {
  return 4ull * max_lod_tris * n_waves > 9ull * total_tris && (unsigned long long)max_lod_tris * command_count > 3ull * total_tris;
}
constexpr uint32_t kPlanTriChoiceWaves = 8192;
constexpr uint32_t kPlanTriBlockTicketsFrom = 32768;  // instances above which the workgroup-per-command kernel pulls tickets
constexpr bool plan_tri_choice_is_block(uint32_t max_lod_tris, uint32_t index_total, uint32_t command_count) {
  const unsigned long long total_tris = (unsigned long long)index_total / 3ull;
  const unsigned long long walk = (unsigned long long)max_lod_tris * kPlanTriChoiceWaves;  // share = walk / total_tris
  return 10ull * walk > 7ull * total_tris || (unsigned long long)max_lod_tris * command_count > 4ull * total_tris;
}

// The whole decision. Order of the checks = order of the error messages round 3's validate_run produced.
inline LaunchPlan plan_frame(const PlanState& st, const PlanRequest& rq) {
  if (!st.have_instances || !st.have_meshes) return plan_refuse(MIP_ERR_NOT_READY, "instances or mesh table not set");
  if (rq.cmds != rq.count) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "draw_cmds and draw_count go together");
  if (rq.index_total && !rq.cmds) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "draw_index_total needs draw_cmds");
  const bool device_out = (rq.flags & MIP_OUT_DEVICE) != 0;
  if ((rq.flags & MIP_OUT_ASYNC) && !device_out) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_ASYNC needs MIP_OUT_DEVICE");
  if (rq.flags & MIP_OUT_WIRE) {
    if (!device_out || !rq.cmds) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE needs MIP_OUT_DEVICE and draw_cmds");
    if (rq.triangles) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE cannot carry the per-triangle stage's indexCount");
    if (rq.skinned) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE is not available for skinned frames");
    if (st.n_meshes > 0x7fffffffu) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE needs mesh ids below 2^31");
    if (rq.cmds_address & 15u) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE needs a 16-byte aligned draw_cmds (block headers are 16-byte stores)");
    if ((rq.flags & MIP_OUT_WIRE_PACKED) && (uint64_t)st.n > (1ull << plan_wire_index_bits(st.n_meshes)))
      return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE_PACKED: the instances do not fit the index bits the mesh table leaves");
  } else if (rq.flags & MIP_OUT_WIRE_PACKED) {
    return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE_PACKED goes with MIP_OUT_WIRE");
  }
  if (rq.triangles) {
    if (!device_out) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "culled_index_buffer needs MIP_OUT_DEVICE");
    if (rq.skinned) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "the per-triangle stage does not skin vertices");
    if (!st.have_geometry) return plan_refuse(MIP_ERR_NOT_READY, "culled_index_buffer needs mip_set_geometry");
    if (!rq.model || !rq.cmds) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "culled_index_buffer needs model and draw_cmds");
  }
  if (rq.tlas && !device_out) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "tlas_instances needs MIP_OUT_DEVICE");
  if (rq.skinned && !device_out) return plan_refuse(MIP_ERR_INVALID_ARGUMENT, "mip_run_skinned needs MIP_OUT_DEVICE outputs");
  if (rq.skinned && (st.n_joints == 0 || st.n_joints > 32u)) return plan_refuse(MIP_ERR_NOT_READY, "skeleton or poses not set for the resident instances");

  LaunchPlan p;
  p.device_out = device_out;
  p.async = device_out && (rq.flags & MIP_OUT_ASYNC) != 0;
  p.need_staging = !device_out;
  p.need_tri_scratch = rq.triangles;
  p.need_skin_box = rq.skinned;
  const uint32_t n = st.n;
  if (n == 0) {
    p.empty = true;
    return p;
  }
  p.n_tiles = plan_tiles_for(n);
  // order (instance_kernel.hpp): commands-first while the launch is ONE generation of workgroups — every tile resident at once
  // (8 workgroups per CU; 5 for the kernel that carries the fall-back arithmetic tiers) —, stores-first beyond. Round 5 measured the
  // crossover point by point (profiles/r05_order_crossover.txt): the commands-first order has a cliff exactly there — 524 288
  // instances = 2 048 tiles 9.95 us, 540 000 12.0 us — while the stores-first order grows smoothly (10.4 -> 11.0), and stays ahead from
  // there on (600 k 11.7 vs 12.8, 786 k 14.6 vs 14.9, 917 k 16.9 vs 17.7). (Rounds 2-4 switched at 14 tiles per CU = 917 k instances.)
  // A frame that streams no per-instance output (no matrices, TLAS rows or boxes: cull + commands only, 41 instead of 105 MB at 1 M)
  // has no store stream for the stores-first order to keep fed, and commands-first stays ahead longer: 1 M 14.4 vs 15.7 us, even at
  // 1.25-1.5 M, behind from 1.75 M (profiles/r04_order_for_command_only_frames.txt).
  const bool streams = rq.model || rq.tlas || rq.aabb;
  const bool general_kernel = rq.skinned || st.nonfinite || st.force_general;
  p.order = p.n_tiles <= st.cu_count * (streams ? (general_kernel ? 5u : 8u) : 17u) ? 3 : 1;
  if (st.force_order == 1 || st.force_order == 3) p.order = st.force_order;
  p.box_override = rq.skinned;
  p.general = rq.skinned || st.nonfinite || st.force_general;  // a per-instance box may be non-finite
  p.wire = device_out ? plan_wire_form(rq.flags) : 0;
  p.group_shift = p.n_tiles <= 512u ? 4u : (p.n_tiles <= 2048u ? 5u : 6u);
  p.uses_prefix_state = rq.cmds;
  if (rq.skinned) {
    p.skin = true;
    const uint32_t per_block = 4u * (64u / st.n_joints);
    p.skin_blocks = (n + per_block - 1u) / per_block;
  }
  if (rq.triangles) {
    // The command count lives on the device; the instance count bounds it.
    //  - tiny frames (the reference's own regime): every command cut into 16 parts, one work item each — only while the
    //    context runs one frame at a time (the grid is sized to be resident as a whole; nothing DEPENDS on that any more,
    //    but two half-resident launches would spend their time helping each other);
    //  - up to tri_block_max instances: one workgroup per command; 1024 / 512 / 256 threads by frame size (measured,
    //    DamagedHelmet entry, frame time in us at 256 / 512 / 1024 threads: 200 instances 35 / 28 / 27, 1000: 60 / 49 / 56,
    //    2000: 72 / 66 / 73, 4000: 85 / 88 / 108, 20 k: 266 / 324 / 347);
    //  - above: one wave per command (100 k: 1.15 vs 1.21 ms), commands pulled from a counter.
    const bool parts = st.tri_parts_max && n <= st.tri_parts_max && !st.tri_block_threads && st.frame_slots == 1 &&
                       st.max_lod_tris <= kPlanTriParts * 256u * kPlanTriPartMaxT;
    if (n >= st.tri_chunks_from && !st.tri_block_threads) {
      // round 5: equal ranges of the triangle stream, one wave each (triangle_kernels.hpp); above tri_block_max instances the same grid
      // takes, per frame and on the device, that decomposition or one wave per command over the size-sorted list (the sort kernels run in front)
      p.tri = n <= st.tri_block_max ? TriangleKernel::chunks : TriangleKernel::sorted;
      p.tri_threads = 256;
      p.need_chunk_scratch = true;
      p.tri_reset_ticket = true;  // long streams: a wave's later ranges are pulled from the counter (and the sorted commands are)
      p.tri_map_blocks = (n + 255u) / 256u;
      p.tri_blocks = st.cu_count * st.tri_chunk_blocks_per_cu;  // resident as a whole: a range's predecessors are running when it looks for them
    } else if (parts) {
      p.tri = TriangleKernel::parts;
      p.tri_threads = 256;
      p.need_part_status = true;
      uint32_t blocks = n * kPlanTriParts;
      const uint32_t max_blocks = st.cu_count * 4u;  // resident as a whole at this kernel's register budget (4 waves per SIMD)
      p.tri_blocks = blocks > max_blocks ? max_blocks : blocks;
    } else if (n <= st.tri_block_max) {
      p.tri = TriangleKernel::block;
      p.tri_threads = st.tri_block_threads ? st.tri_block_threads : (n <= 768u ? 1024u : (n <= 3072u ? 512u : 256u));
      const uint32_t per_cu = 2u * (1024u / p.tri_threads);
      uint32_t blocks = n < st.cu_count * per_cu ? n : st.cu_count * per_cu;
      p.tri_blocks = blocks ? blocks : 1u;
      // many more commands than workgroups: pulled from the counter, not dealt by a stride (one-mesh scene 20 k instances
      // 0.24 vs 0.25 ms for the stride, 40 k 0.48 -> 0.45; mixed 60 k 0.34 -> 0.30: profiles/r04_triangle_block_tickets.txt)
      p.tri_block_tickets = p.tri_reset_ticket = n > kPlanTriBlockTicketsFrom;
    } else {
      p.tri = TriangleKernel::waves;
      p.tri_threads = 256;
      p.tri_reset_ticket = true;
      uint32_t blocks = (n + 3u) / 4u;
      const uint32_t max_blocks = st.cu_count * 8u;
      p.tri_blocks = blocks > max_blocks ? max_blocks : blocks;
      // Which of the two is faster depends on the frame, not on the instance count: a wave walks a command alone, so the
      // launch ends with the largest commands' tails (mixed scene, 100 k instances: 0.72 ms against 0.49 for a workgroup per
      // command; 400 k: equal; one-mesh scene: the wave kernel 6 % ahead). The totals that decide it are on the device when
      // the stage starts, so BOTH grids are launched and one of them returns at once (tri_choice).
      if (!st.tri_no_choice) p.tri_either_blocks = st.cu_count * 8u;
    }
    if (n <= st.tri_block_max) {
      p.recompact = Recompact::single;
      p.recompact_blocks = 1;
    } else {  // many commands: counts per 1024, one block scans them, scatter
      p.recompact = Recompact::wide;
      p.recompact_blocks = (n + 1023u) / 1024u;
    }
  }
  return p;
}

}  // namespace mip
