// frame_plan_check.cpp — enumerates plan_frame (renderer_amd/csrc/frame_plan.hpp) over every combination of context state and
// requested outputs and checks the invariants the launch code (api_frame.hip) relies on. Plain C++, no HIP: built by
// tests/test_frame_plan.py with gcc -fsanitize=address,undefined. Prints one line per property and "PLAN OK <combinations>".
#include "../../renderer_amd/csrc/frame_plan.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>

using namespace mip;

static unsigned long long combos = 0, refused = 0, planned = 0;

#define CHECK(cond, ...)                                                        \
  do {                                                                          \
    if (!(cond)) {                                                              \
      std::printf("FAILED %s:%d %s — ", __FILE__, __LINE__, #cond);             \
      std::printf(__VA_ARGS__);                                                 \
      std::printf("\n");                                                        \
      std::exit(1);                                                             \
    }                                                                           \
  } while (0)

static void check_one(const PlanState& st, const PlanRequest& rq) {
  ++combos;
  const LaunchPlan p = plan_frame(st, rq);
  const bool device_out = (rq.flags & MIP_OUT_DEVICE) != 0;
  if (p.status != MIP_OK) {
    ++refused;
    CHECK(p.why && std::strlen(p.why) > 8, "a refusal says why");
    CHECK(p.status == MIP_ERR_INVALID_ARGUMENT || p.status == MIP_ERR_NOT_READY, "status %d", p.status);
    // nothing that is valid may be refused: re-derive validity independently
    const bool wire = (rq.flags & MIP_OUT_WIRE) != 0, packed = (rq.flags & MIP_OUT_WIRE_PACKED) != 0;
    bool valid = st.have_instances && st.have_meshes && rq.cmds == rq.count && (!rq.index_total || rq.cmds) &&
                 (!(rq.flags & MIP_OUT_ASYNC) || device_out) && (!packed || wire);
    if (wire) valid = valid && device_out && rq.cmds && !rq.triangles && !rq.skinned && !(rq.cmds_address & 15u) && st.n_meshes <= 0x7fffffffu &&
                      (!packed || (uint64_t)st.n <= (1ull << plan_wire_index_bits(st.n_meshes)));
    if (rq.triangles) valid = valid && device_out && !rq.skinned && st.have_geometry && rq.model && rq.cmds;
    if (rq.tlas) valid = valid && device_out;
    if (rq.skinned) valid = valid && device_out && st.n_joints >= 1 && st.n_joints <= 32;
    CHECK(!valid, "a valid request was refused: %s", p.why);
    return;
  }
  ++planned;
  CHECK(p.device_out == device_out, "device_out");
  CHECK(p.async == (device_out && (rq.flags & MIP_OUT_ASYNC)), "async only with device outputs");
  CHECK(p.need_staging == !device_out, "host outputs need staging, device outputs never");
  if (st.n == 0) {
    CHECK(p.empty && p.n_tiles == 0 && p.tri == TriangleKernel::none && !p.skin, "an empty scene launches nothing");
    return;
  }
  CHECK(!p.empty, "resident instances: something launches");
  CHECK(p.n_tiles == (st.n + 255u) / 256u, "one tile per 256 instances");
  CHECK(p.order == 1 || p.order == 3, "order %d", p.order);
  if (st.force_order) CHECK(p.order == st.force_order, "forced order");
  else CHECK((p.order == 3) == (p.n_tiles <= st.cu_count * ((rq.model || rq.tlas || rq.aabb) ? (p.general ? 5u : 8u) : 17u)),
             "order: commands-first while the launch is one generation of workgroups (or, without a per-instance stream, up to 17 tiles per CU)");
  CHECK(p.group_shift >= 4 && p.group_shift <= 6 && ((p.n_tiles + (1u << p.group_shift) - 1u) >> p.group_shift) <= (st.max_instances / 256u + 16u) / 16u + 1u,
        "groups fit the prefix state sized at create (smallest group: 16 tiles)");
  CHECK(p.box_override == rq.skinned, "box override <=> skinned");
  CHECK(!p.box_override || p.general, "a box override may be non-finite: general kernel");
  CHECK(p.general == (rq.skinned || st.nonfinite || st.force_general), "general kernel exactly when needed");
  CHECK(p.wire == (device_out ? plan_wire_form(rq.flags) : 0), "wire form");
  CHECK(!(p.wire && p.box_override), "no kernel instantiation has wire + box override");
  CHECK(!(p.wire && rq.triangles), "the wire form never feeds the triangle stage");
  CHECK(p.uses_prefix_state == rq.cmds, "a tag is spent exactly when commands are emitted");
  CHECK(p.skin == rq.skinned && p.need_skin_box == rq.skinned, "skinning kernel <=> skinned frame");
  if (p.skin) {
    const uint32_t per_block = 4u * (64u / st.n_joints);
    CHECK((uint64_t)p.skin_blocks * per_block >= st.n && (uint64_t)(p.skin_blocks - 1u) * per_block < st.n, "skin grid covers the instances exactly");
  }
  CHECK((p.tri != TriangleKernel::none) == rq.triangles && p.need_tri_scratch == rq.triangles, "triangle stage <=> culled_index_buffer");
  CHECK((p.recompact != Recompact::none) == rq.triangles, "re-compaction follows the triangle stage");
  if (rq.triangles) {
    CHECK(p.tri_blocks >= 1, "a triangle launch has a grid");
    CHECK(p.need_part_status == (p.tri == TriangleKernel::parts), "granules only for the parts kernel");
    CHECK(p.need_chunk_scratch == (p.tri == TriangleKernel::chunks || p.tri == TriangleKernel::sorted), "range map and granules only for the range kernel");
    CHECK(p.tri_reset_ticket == (p.tri == TriangleKernel::waves || p.tri == TriangleKernel::chunks || p.tri == TriangleKernel::sorted || p.tri_block_tickets), "the counter is zeroed exactly for the kernels that pull from it");
    CHECK(p.tri_block_tickets == (p.tri == TriangleKernel::block && st.n > 32768u), "the workgroup kernel pulls tickets when commands far outnumber workgroups");
    if (p.tri == TriangleKernel::chunks || p.tri == TriangleKernel::sorted) {
      CHECK(st.n >= st.tri_chunks_from && !st.tri_block_threads, "range kernel preconditions");
      CHECK((p.tri == TriangleKernel::sorted) == (st.n > st.tri_block_max), "sorted commands + device-side choice above tri_block_max");
      CHECK(p.tri_blocks == st.cu_count * st.tri_chunk_blocks_per_cu && p.tri_threads == 256, "chunk grid: resident as a whole");
      CHECK((uint64_t)p.tri_map_blocks * 256u >= st.n && (uint64_t)(p.tri_map_blocks - 1u) * 256u < st.n, "chunk map: one thread per possible command");
    } else if (p.tri == TriangleKernel::parts) {
      CHECK(st.n < st.tri_chunks_from || st.tri_block_threads, "the chunk kernel comes first where it is switched on");
      CHECK(st.frame_slots == 1 && st.n <= st.tri_parts_max && !st.tri_block_threads && st.max_lod_tris <= 16u * 256u * 8u, "parts kernel preconditions");
      CHECK(p.tri_blocks <= st.cu_count * 4u && p.tri_blocks <= st.n * 16u, "parts grid: resident as a whole, no more blocks than items");
    } else if (p.tri == TriangleKernel::block) {
      CHECK(st.n <= st.tri_block_max, "workgroup-per-command only up to tri_block_max");
      CHECK(p.tri_threads == 256 || p.tri_threads == 512 || p.tri_threads == 1024, "block threads %u", p.tri_threads);
      CHECK(p.tri_blocks <= st.n && p.tri_blocks <= st.cu_count * 2u * (1024u / p.tri_threads), "block grid");
    } else {
      CHECK(st.n > st.tri_block_max, "wave-per-command only above tri_block_max");
      CHECK(p.tri_blocks <= st.cu_count * 8u && (uint64_t)p.tri_blocks * 4u >= (st.n < st.cu_count * 32u ? st.n : 1u), "waves grid");
      CHECK(p.tri_either_blocks == (st.tri_no_choice ? 0u : st.cu_count * 8u), "large frames launch both grids unless the choice is tuned off");
    }
    if (p.tri != TriangleKernel::waves) CHECK(p.tri_either_blocks == 0, "only large frames choose on the device");
    CHECK((p.recompact == Recompact::single) == (st.n <= st.tri_block_max), "one-workgroup re-compaction for small frames");
    if (p.recompact == Recompact::wide) CHECK((uint64_t)p.recompact_blocks * 1024u >= st.n, "wide re-compaction covers the list");
  } else {
    CHECK(!p.need_part_status && !p.need_chunk_scratch && !p.tri_reset_ticket && !p.tri_map_blocks, "no triangle scratch without the stage");
  }
}

// The device-side choice between the two large-frame triangle grids, on the frames it was measured on
// (profiles/r04_triangle_kernel_choice.txt): {largest command, total indices, commands} -> workgroup-per-command grid?
static void check_tri_choice() {
  struct Row { const char* what; uint32_t max_lod_tris, index_total, commands; bool block; };
  const Row rows[] = {
      {"mixed 70 k", 23358, 41000000u * 3u, 18220, true},     {"mixed 100 k", 23358, 57600000u * 3u, 25776, true},
      {"mixed 200 k", 23358, 114400000u * 3u, 51355, true},   {"mixed 400 k", 23358, 229300000u * 3u, 103015, true},
      {"mixed 600 k", 23358, 343900000u * 3u, 154705, true},  {"mixed 1 M", 23358, 571400000u * 3u, 257864, true},
      {"one mesh 70 k", 15452, 149500000u * 3u, 18353, true}, {"one mesh 100 k", 15452, 213000000u * 3u, 26154, false},
      {"one mesh 150 k", 15452, 319000000u * 3u, 39209, false}, {"one mesh 300 k", 15452, 638000000u * 3u, 78400, false},
      {"an empty frame", 15452, 0u, 0u, true},                {"one command", 15452, 15452u * 3u, 1u, true}};
  for (const Row& r : rows)
    CHECK(plan_tri_choice_is_block(r.max_lod_tris, r.index_total, r.commands) == r.block, "triangle grid choice for %s", r.what);
  // round 5: range kernel or size-sorted wave-per-command kernel, on the frames it was measured on (8 192 waves)
  struct Row5 { const char* what; uint32_t max_lod_tris, total_tris, commands; bool ranges; };
  const Row5 rows5[] = {
      {"mixed 70 k", 23358, 41000000u, 18220, true},      {"mixed 100 k", 23358, 57600000u, 25776, true},   {"mixed 130 k", 23358, 74900000u, 33500, true},
      {"mixed 160 k", 23358, 92200000u, 41200, false},    {"mixed 200 k", 23358, 114400000u, 51355, false},
      {"mixed 400 k", 23358, 229300000u, 103015, false},  {"mixed 1 M", 23358, 571400000u, 257864, false},  {"one mesh 70 k", 15452, 149500000u, 18353, false},
      {"one mesh 100 k", 15452, 213000000u, 26154, false}, {"one mesh 300 k", 15452, 638000000u, 78400, false}};
  for (const Row5& r : rows5)
    CHECK(plan_tri_choice_is_ranges(r.max_lod_tris, r.total_tris, r.commands, 8192u) == r.ranges, "range / sorted-wave choice for %s", r.what);
  // monotone: more triangles in the frame never turn the wave-per-command choice back into the workgroup one
  for (uint32_t max_tris : {12u, 1000u, 23358u})
    for (uint32_t commands : {1u, 1000u, 100000u}) {
      bool seen_waves = false;
      for (uint64_t total = 0; total <= 0xffffffffull; total = total * 2 + 3) {
        const bool block = plan_tri_choice_is_block(max_tris, (uint32_t)total, commands);
        if (!block) seen_waves = true;
        CHECK(!(seen_waves && block), "choice not monotone in the frame's total (max %u, %u commands, total %llu)", max_tris, commands, (unsigned long long)total);
      }
    }
}

int main() {
  check_tri_choice();
  const uint32_t sizes[] = {0u, 1u, 257u, 768u, 1025u, 3073u, 65537u, 327680u, 327681u, 524288u, 524289u, 1114112u, 1114113u, 0x3fffffffu};  // the order thresholds at 256 CUs
  const uint32_t flag_sets[] = {MIP_OUT_HOST, MIP_OUT_DEVICE, MIP_OUT_DEVICE | MIP_OUT_ASYNC, MIP_OUT_ASYNC, MIP_OUT_DEVICE | MIP_OUT_WIRE,
                                MIP_OUT_DEVICE | MIP_OUT_WIRE | MIP_OUT_WIRE_PACKED, MIP_OUT_DEVICE | MIP_OUT_WIRE_PACKED, MIP_OUT_WIRE,
                                MIP_OUT_DEVICE | MIP_OUT_ASYNC | MIP_OUT_WIRE | MIP_OUT_WIRE_PACKED};
  for (uint32_t n : sizes)
    for (uint32_t n_meshes : {64u, 0x80000000u})
      for (uint32_t slots : {1u, 2u})
        for (int state_bits = 0; state_bits < 32; ++state_bits)
          for (int force_order : {0, 1})
            for (uint32_t tri_threads : {0u, 512u})
              for (uint32_t n_joints : {0u, 19u})
                for (uint32_t lod_tris : {32768u, 32769u})
                 for (uint32_t chunks_from : {0u, 2000u, 0xffffffffu}) {
                  PlanState st;
                  st.tri_chunks_from = chunks_from;
                  st.n = n;
                  st.n_meshes = n_meshes;
                  st.max_instances = n ? n : 1u;
                  st.frame_slots = slots;
                  st.have_instances = state_bits & 1;
                  st.have_meshes = state_bits & 2;
                  st.have_geometry = state_bits & 4;
                  st.nonfinite = state_bits & 8;
                  st.force_general = state_bits & 16;
                  st.force_order = force_order;
                  st.tri_block_threads = tri_threads;
                  st.n_joints = n_joints;
                  st.max_lod_tris = lod_tris;
                  for (uint32_t flags : flag_sets)
                    for (int out_bits = 0; out_bits < 512; ++out_bits) {
                      PlanRequest rq;
                      rq.model = out_bits & 1;
                      rq.bitmap = out_bits & 2;
                      rq.cmds = out_bits & 4;
                      rq.count = out_bits & 8;
                      rq.index_total = out_bits & 16;
                      rq.aabb = out_bits & 32;
                      rq.tlas = out_bits & 64;
                      rq.triangles = out_bits & 128;
                      rq.skinned = out_bits & 256;
                      rq.flags = flags;
                      rq.cmds_address = rq.cmds ? 0x7f0000001000ull : 0ull;
                      check_one(st, rq);
                      if (rq.cmds && (flags & MIP_OUT_WIRE)) {  // the same with a list that is only 4-byte aligned
                        rq.cmds_address += 4;
                        check_one(st, rq);
                      }
                    }
                }
  // a few knobs that the sweep above leaves at their defaults
  for (uint32_t parts_max : {0u, 1024u, 5000u})
    for (uint32_t block_max : {0u, 65536u, 2000000u})
      for (uint32_t cu : {1u, 64u, 256u, 304u})
        for (uint32_t n : sizes) {
          PlanState st;
          st.n = n; st.n_meshes = 3; st.max_instances = n ? n : 1u; st.cu_count = cu;
          st.have_instances = st.have_meshes = st.have_geometry = true;
          st.tri_parts_max = parts_max; st.tri_block_max = block_max; st.max_lod_tris = 1000;
          PlanRequest rq;
          rq.model = rq.cmds = rq.count = rq.triangles = true;
          rq.flags = MIP_OUT_DEVICE;
          rq.cmds_address = 0x1000;
          for (bool no_choice : {false, true}) {
            st.tri_no_choice = no_choice;
            check_one(st, rq);
          }
        }
  CHECK(plan_wire_index_bits(0) == 31 && plan_wire_index_bits(1) == 31 && plan_wire_index_bits(2) == 30 && plan_wire_index_bits(64) == 25 &&
        plan_wire_index_bits(65) == 24 && plan_wire_index_bits(0xffffffffu) == 0, "index bits of a packed record");
  std::printf("PLAN OK %llu combinations (%llu planned, %llu refused)\n", combos, planned, refused);
  return 0;
}
