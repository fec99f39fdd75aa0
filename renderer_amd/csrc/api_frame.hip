// api_frame.hip — C ABI of the instance pipeline, part 2 of 4: one frame. plan_frame (frame_plan.hpp) decides which
// kernels run; this file executes that plan: tags of the cross-tile prefix state, the launches, the copy-back of host
// outputs, recorded launch graphs (mip_run_many), the multi-view launch, the shadow-pass lists, mip_wait.
// The stores-first order of the frame kernel (kOrder == 1) is instantiated HERE and only here; the commands-first
// order lives in stages_tu.hip, which is built with other flags (stage_args.hpp).
#include "context.hpp"

namespace mip_host {
namespace {

int32_t ensure_staging(MipContext* ctx, const MipOutputs* out) {
  const size_t cap = ctx->max_instances ? ctx->max_instances : 1;
  if (out->model && !ctx->s_model) MIP_HIP(ctx, hipMalloc(&ctx->s_model, cap * 64));
  if (out->visible_bitmap && !ctx->s_bitmap) MIP_HIP(ctx, hipMalloc(&ctx->s_bitmap, ((cap + 31) / 32) * 4));
  if (out->draw_cmds && !ctx->s_cmds) MIP_HIP(ctx, hipMalloc(&ctx->s_cmds, cap * 20));
  if (out->world_aabb && !ctx->s_aabb) MIP_HIP(ctx, hipMalloc(&ctx->s_aabb, cap * 24));
  return MIP_OK;
}

using FrameKernel = mip::FrameKernelFn;
template <bool kBox, bool kGeneral, int kWire>
FrameKernel pick_order(int order, bool first_mover) {
  if (order != 1) return mip::frame_kernel_commands_first(kBox, kGeneral, kWire, first_mover);
  if constexpr (mip::KernelArgs::kFirstMoverAdds)
    if (first_mover) return (FrameKernel)mip::mip_instance_pipeline_kernel<kBox, kGeneral, 1, kWire, true>;
  return (FrameKernel)mip::mip_instance_pipeline_kernel<kBox, kGeneral, 1, kWire>;
}
// The instantiation a plan names; `first_mover`: the one that follows the first-mover rule (KernelArgs.first_mover_rule == 1).
FrameKernel frame_kernel_of(const mip::LaunchPlan& p, bool first_mover) {
  if (p.box_override) return pick_order<true, true, 0>(p.order, first_mover);  // (plan_frame refuses MIP_OUT_WIRE for skinned frames)
  if (p.wire == 2) return p.general ? pick_order<false, true, 2>(p.order, first_mover) : pick_order<false, false, 2>(p.order, first_mover);
  if (p.wire == 1) return p.general ? pick_order<false, true, 1>(p.order, first_mover) : pick_order<false, false, 1>(p.order, first_mover);
  return p.general ? pick_order<false, true, 0>(p.order, first_mover) : pick_order<false, false, 0>(p.order, first_mover);
}

mip::PlanState plan_state(const MipContext* ctx) {
  mip::PlanState st;
  st.n = ctx->n;
  st.n_meshes = ctx->m;
  st.max_instances = ctx->max_instances;
  st.cu_count = (uint32_t)ctx->cu_count;
  st.frame_slots = (uint32_t)ctx->slots.size();
  st.have_instances = ctx->have_instances;
  st.have_meshes = ctx->have_meshes;
  st.have_geometry = ctx->have_geometry;
  st.nonfinite = ctx->nonfinite_instances != 0;
  st.force_general = ctx->force_general;
  st.force_order = ctx->force_order;
  st.tri_block_threads = ctx->tri_block_threads;
  st.tri_block_max = ctx->tri_block_max;
  st.tri_parts_max = ctx->tri_parts_max;
  st.tri_chunks_from = ctx->tri_chunks_from;
  st.tri_chunk_blocks_per_cu = mip::triangle_chunks_blocks_per_cu();
  st.tri_no_choice = ctx->tri_no_choice;
  st.max_lod_tris = ctx->max_lod_tris;
  st.n_joints = ctx->n_joints;
  return st;
}

mip::PlanRequest plan_request(const MipOutputs* out, bool skinned) {
  mip::PlanRequest rq;
  rq.model = out->model != nullptr;
  rq.bitmap = out->visible_bitmap != nullptr;
  rq.cmds = out->draw_cmds != nullptr;
  rq.count = out->draw_count != nullptr;
  rq.index_total = out->draw_index_total != nullptr;
  rq.aabb = out->world_aabb != nullptr;
  rq.tlas = out->tlas_instances != nullptr;
  rq.triangles = out->culled_index_buffer != nullptr;
  rq.skinned = skinned;
  rq.flags = out->flags;
  rq.cmds_address = (uintptr_t)out->draw_cmds;
  return rq;
}

// Whether the launch that is being prepared follows the frame kernel's first-mover rule: for kFirstMoverLaunches launches after
// tile 0 of some launch found that waves had to help (the device writes the help count to a pinned word; reading it costs nothing).
uint32_t first_mover_rule_now(MipContext* ctx) {
  if (ctx->first_mover_env) return ctx->first_mover_env == 1u ? 1u : 0u;
  const uint32_t hint = ((volatile uint32_t*)ctx->h_error)[kHelpHintWord];
  if (hint != ctx->help_hint_seen) {
    ctx->help_hint_seen = hint;
    ctx->first_mover_launches_left = kFirstMoverLaunches;
  }
  if (ctx->first_mover_launches_left == 0) return 0u;
  ctx->first_mover_launches_left -= 1;
  return 1u;
}

// Everything of a launch except the tag: resident inputs, output pointers, prefix state, frame.
void fill_kernel_args(MipContext* ctx, MipContext::FrameSlot& sl, const MipFrame* frame, const MipOutputs* out,
                      bool device_out, mip::KernelArgs& a) {
  const uint32_t n = ctx->n;
  a.pos = ctx->d_pos; a.rot = ctx->d_rot; a.scale = ctx->d_scale; a.mesh_id = ctx->d_mesh_id;
  a.meshes = ctx->d_meshes; a.mesh_draw = ctx->d_mesh_draw;
  a.model = out->model ? (device_out ? (float4*)out->model : ctx->s_model) : nullptr;
  a.bitmap = out->visible_bitmap ? (device_out ? out->visible_bitmap : ctx->s_bitmap) : nullptr;
  a.cmds = out->draw_cmds ? (device_out ? (uint32_t*)out->draw_cmds : ctx->s_cmds) : nullptr;
  a.draw_count = out->draw_cmds ? (device_out ? out->draw_count : sl.d_scalars + 0) : nullptr;
  a.index_total = out->draw_cmds ? ((device_out && out->draw_index_total) ? out->draw_index_total : sl.d_scalars + 1) : nullptr;
  a.world_aabb = out->world_aabb ? (device_out ? (float*)out->world_aabb : ctx->s_aabb) : nullptr;
  if (out->tlas_instances && device_out) {
    a.tlas_instances = (uint4*)out->tlas_instances;
    a.blas_address = ctx->d_blas;
  }
  a.status0 = sl.d_status;
  a.acc1 = sl.d_status + ctx->acc1_offset_words;
  a.start1 = sl.d_status + ctx->start1_offset_words;
  a.groups_cap = ctx->groups_cap;
  a.error_flag = ctx->d_error;
  a.help_counter = ctx->d_help;
  a.helps_seen = reinterpret_cast<uint32_t*>(sl.d_status + ctx->helps_seen_offset_words);
  a.help_hint = ctx->first_mover_env ? nullptr : ctx->d_error + kHelpHintWord;
  a.first_mover_rule = ctx->first_mover_env == 1u ? 1u : 0u;  // (recorded launches keep this; a direct launch asks first_mover_rule_now)
  a.n = n;
  a.bitmap_words = (n + 31u) / 32u;
  a.n_meshes = ctx->m;
  a.one_mesh = ctx->m == 1u && !ctx->no_one_mesh ? 1u : 0u;
  a.wire_index_bits = mip_wire_index_bits(ctx->m);
  a.first_instance_base = frame->first_instance_base;
  a.first_index_base = frame->first_index_base;
  std::memcpy(a.planes, frame->planes, sizeof a.planes);
  std::memcpy(a.cam, frame->cam_pos, sizeof a.cam);
  a.n_tiles = tiles_for(n);
#ifdef MIP_EXP_FAKE_DELAY
  a.delay_first = std::getenv("MIP_TUNE_DELAY_FIRST") ? (uint32_t)std::atoi(std::getenv("MIP_TUNE_DELAY_FIRST")) : 0u;
  a.delay_last = std::getenv("MIP_TUNE_DELAY_LAST") ? (uint32_t)std::atoi(std::getenv("MIP_TUNE_DELAY_LAST")) : 0xffffffffu;
#endif
  a.group_shift = a.n_tiles <= 512 ? 4u : (a.n_tiles <= 2048 ? 5u : 6u);
#ifdef MIP_DEBUG_STAMPS
  a.stamps = ctx->d_stamps;
  // fault injection (diagnostic build): a tile that never publishes; a permutation of the tile numbers (workgroups then
  // start in an order that is anything but ascending) — the frame must come out byte-identical either way
  if (const char* env = std::getenv("MIP_DEBUG_SKIP_PUBLISH_TILE")) a.debug_skip_publish_tile = (uint32_t)std::atoi(env) + 1u;
  if (const char* env = std::getenv("MIP_DEBUG_TILE_ORDER")) {
    const uint32_t t = a.n_tiles;
    if (t > 1u && std::strcmp(env, "reverse") == 0) {
      a.debug_tile_mult = t - 1u;
      a.debug_tile_add = t - 1u;
    } else if (t > 1u && std::strcmp(env, "scramble") == 0) {
      static const uint32_t primes[] = {7919u, 104729u, 1299709u, 15485863u};
      for (uint32_t p : primes)
        if (t % p != 0u) { a.debug_tile_mult = p; break; }  // a prime that does not divide t is coprime to it
      a.debug_tile_add = 12345u % t;
    }
  }
#endif
}

// Clears a slot's prefix state when the instance count changed or fewer than `need` tags are left.
int32_t reset_prefix_state_if_needed(MipContext* ctx, MipContext::FrameSlot& sl, uint32_t need) {
  if (sl.status_dirty || sl.epoch + need > mip::kMaxEpoch) {
    MIP_HIP(ctx, hipMemsetAsync(sl.d_status, 0, ctx->status_bytes, sl.stream));
    sl.status_dirty = false;
    sl.epoch = sl.last_tag = 0;
    sl.zero_buf = 2;
    ctx->graph_generation++;  // recorded tags are meaningless on a cleared state
  }
  return MIP_OK;
}

// One pass over the LOD ranges the per-triangle stage can be sent to (LODs 0 and 1 of every mesh: pick_lod
// never picks another): each must lie inside the uploaded index buffer, and every index in it, offset by the
// mesh's vertex_offset, inside the uploaded vertices. Cached until either table changes.
int32_t check_geometry(MipContext* ctx) {
  if (ctx->geometry_checked == 0) {
    ctx->geometry_checked = 1;
    char buf[256];
    for (uint32_t k = 0; k < ctx->h_meshes.size() && ctx->geometry_checked == 1; ++k) {
      const MipMesh& m = ctx->h_meshes[k];
      for (uint32_t l = 0; l < m.n_lods && l < 2u; ++l) {
        const uint64_t off = m.index_offset[l], len = m.index_len[l];
        if (off + len > ctx->n_indices) {
          snprintf(buf, sizeof buf, "mesh %u LOD %u: indices [%llu, %llu) outside the %u uploaded indices", k, l,
                   (unsigned long long)off, (unsigned long long)(off + len), ctx->n_indices);
          ctx->geometry_error = buf;
          ctx->geometry_checked = -1;
          break;
        }
        uint32_t mx = 0;
        for (uint64_t j = off; j < off + len; ++j) mx = ctx->h_indices[j] > mx ? ctx->h_indices[j] : mx;
        if (len && (m.vertex_offset < 0 || (uint64_t)m.vertex_offset + mx >= ctx->n_vertices)) {
          snprintf(buf, sizeof buf, "mesh %u LOD %u: vertex_offset %d + largest index %u outside the %u uploaded vertices", k, l,
                   m.vertex_offset, mx, ctx->n_vertices);
          ctx->geometry_error = buf;
          ctx->geometry_checked = -1;
          break;
        }
      }
    }
  }
  if (ctx->geometry_checked < 0)
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "mesh table and geometry disagree: %s", ctx->geometry_error.c_str());
  return MIP_OK;
}

// plan_frame + the one check that needs the context's host copies (mesh table against geometry).
int32_t plan_for(MipContext* ctx, const MipFrame* frame, const MipOutputs* out, bool skinned, mip::LaunchPlan* plan) {
  if (!frame || !out) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "frame/out is NULL");
  *plan = mip::plan_frame(plan_state(ctx), plan_request(out, skinned));
  if (plan->status != MIP_OK)
    return fail(ctx, plan->status, "%s (%u instances, %u meshes, output flags 0x%x)", plan->why, ctx->n, ctx->m, out->flags);
  if (out->culled_index_buffer)
    if (int32_t rc = check_geometry(ctx)) return rc;
  return MIP_OK;
}

}  // namespace

void drop_graphs(MipContext* ctx) {
  for (auto& gs : ctx->graph_sets)
    for (auto& fg : gs.per_slot) {
      if (fg.exec) (void)hipGraphExecDestroy(fg.exec);
      if (fg.graph) (void)hipGraphDestroy(fg.graph);
    }
  ctx->graph_sets.clear();
}

// One frame on the next frame slot. Every decision is in `plan` (frame_plan.hpp); what is left here is the order of
// the launches and the bookkeeping of the slot's prefix state.
int32_t run_frame(MipContext* ctx, const MipFrame* frame, const MipOutputs* out, bool skinned, void* palette) {
  mip::LaunchPlan plan;
  if (int32_t rc = plan_for(ctx, frame, out, skinned, &plan)) return rc;
  if (int32_t rc = bind_device(ctx)) return rc;
  const bool device_out = plan.device_out, async = plan.async;

  // Frames rotate over the slots; a slot's stream orders a frame after the frame that last
  // used the same prefix state.
  MipContext::FrameSlot& sl = ctx->slots[ctx->next_slot];
  ctx->last_slot = ctx->next_slot;
  ctx->next_slot = (ctx->next_slot + 1) % (uint32_t)ctx->slots.size();
  hipStream_t stream = sl.stream;

  const uint32_t n = ctx->n;
  const uint32_t words = (n + 31u) / 32u;
  const size_t cap = ctx->max_instances ? ctx->max_instances : 1;
  if (plan.need_staging)
    if (int32_t rc = ensure_staging(ctx, out)) return rc;

  mip::KernelArgs a{};
  fill_kernel_args(ctx, sl, frame, out, device_out, a);
  a.group_shift = plan.group_shift;
  if (plan.uses_prefix_state) a.first_mover_rule = first_mover_rule_now(ctx);
  if (plan.need_tri_scratch) {
    // the instance kernel emits into the slot's scratch list; the triangle stage rewrites
    // indexCount there and the final compaction lands in the caller's buffers
    if (!sl.d_tmp_cmds) MIP_HIP(ctx, hipMalloc(&sl.d_tmp_cmds, cap * 20));
    if (!sl.d_tmp_src) MIP_HIP(ctx, hipMalloc(&sl.d_tmp_src, cap * 4));
    if (!sl.d_tmp_blocks) {  // one granule per 1024 commands (the one-launch re-compaction; the round-4 form uses the first half as words)
      MIP_HIP(ctx, hipMalloc(&sl.d_tmp_blocks, (cap / 1024 + 1) * 8));
      MIP_HIP(ctx, hipMemsetAsync(sl.d_tmp_blocks, 0, (cap / 1024 + 1) * 8, stream));
      sl.recompact_epoch = 0;
    }
    if (!sl.d_tmp_final) MIP_HIP(ctx, hipMalloc(&sl.d_tmp_final, cap * 4));
    a.cmds = sl.d_tmp_cmds;
    a.draw_count = sl.d_scalars + 2;
    a.src_index_offset = sl.d_tmp_src;
  }
  if (plan.need_skin_box) {
    // the posed mesh-space box replaces the mesh table's; computed first, on the same stream
    if (!sl.d_skin_box) MIP_HIP(ctx, hipMalloc(&sl.d_skin_box, cap * 32));
    a.box_override = sl.d_skin_box;  // per frame slot: frames in flight may carry different poses
  }

  const bool timing = (ctx->cfg_flags & MIP_CFG_TIMING) != 0;
  if (plan.empty) {
    if (out->culled_index_buffer) MIP_HIP(ctx, hipMemsetAsync(out->draw_count, 0, 4, stream));
    if (a.draw_count) MIP_HIP(ctx, hipMemsetAsync(a.draw_count, 0, 4, stream));
    if (a.index_total) MIP_HIP(ctx, hipMemsetAsync(a.index_total, 0, 4, stream));
  } else {
    // Cross-tile prefix state (see instance_kernel.hpp): a fresh tag per launch marks
    // the level-0 words; the level-1 accumulators alternate between two buffers by tag parity,
    // the kernel zeroing the other one. Launches without draw commands do not touch the state.
    if (plan.uses_prefix_state) {
      if (int32_t rc = reset_prefix_state_if_needed(ctx, sl, 2)) return rc;
      uint32_t e = sl.epoch + 1;
      if (sl.zero_buf != 2 && (e & 1u) != sl.zero_buf) ++e;  // must accumulate in the zeroed buffer
      a.epoch = sl.epoch = sl.last_tag = e;
      sl.zero_buf = (e & 1u) ^ 1u;
    }
    if (timing) MIP_HIP(ctx, hipEventRecord(ctx->ev0, stream));
    if (plan.skin) {
      mip::SkinArgs k{};
      k.poses = ctx->d_poses;
      k.joints = ctx->d_joints;
      k.palette = (float4*)palette;
      k.local_box = sl.d_skin_box;
      k.n = n;
      k.n_joints = ctx->n_joints;
      k.max_depth = ctx->max_joint_depth;
      k.box_bound = ctx->joint_box_bound;
      k.inv_joints = (65536u + ctx->n_joints - 1u) / ctx->n_joints;
      std::memcpy(k.level_start, ctx->joint_level_start, sizeof k.level_start);
      std::memcpy(k.level_inv, ctx->joint_level_inv, sizeof k.level_inv);
      mip::launch_skinned_bounds(plan.skin_blocks, stream, k);
      MIP_HIP(ctx, hipGetLastError());
    }
    {
      mip::FrameKernelParams params(a);
      if (plan.general) ctx->timings.general_launches += 1;
      MIP_HIP(ctx, hipLaunchKernel((const void*)frame_kernel_of(plan, a.first_mover_rule == 1u), dim3(plan.n_tiles), dim3(mip::kTile), params.p, ctx->lds_pad, stream));
    }
    if (plan.tri != mip::TriangleKernel::none) {
      mip::TriangleArgs t{};
      t.cmds = sl.d_tmp_cmds;
      t.count = sl.d_scalars + 2;
      t.src_index_offset = sl.d_tmp_src;
      t.model = (const float4*)out->model;
      t.vertices = ctx->d_vertices;
      t.vertex_bytes = (unsigned long long)ctx->n_vertices * 12ull;
      t.indices = ctx->d_indices;
      t.out_indices = (uint32_t*)out->culled_index_buffer;
      t.capacity = out->culled_index_capacity;
      t.first_instance_base = frame->first_instance_base;
      t.error_flag = ctx->d_error;
      t.help_counter = ctx->d_help + mip::kHelpShards;
      t.ticket = sl.d_scalars + 3;
      t.geometry_finite = ctx->geometry_finite ? 1u : 0u;
      std::memcpy(t.pv, frame->pv, sizeof t.pv);
      uint32_t* zero_words = nullptr;  // counters of the stage that the re-compaction clears for the slot's next frame
      uint32_t n_zero = 0;
      if (plan.tri != mip::TriangleKernel::chunks) sl.tri_ticket_clean = false;  // (the round-4 kernels leave the shared ticket word as it ends)
      if (plan.tri == mip::TriangleKernel::chunks || plan.tri == mip::TriangleKernel::sorted) {
        const bool either = plan.tri == mip::TriangleKernel::sorted;  // one grid, either decomposition, chosen on the device (mip_triangle_stage_kernel)
        t.first_index_base = frame->first_index_base;
        t.max_lod_tris = ctx->max_lod_tris;
        if (ctx->tri_force_choice) t.max_lod_tris = ctx->tri_force_choice == 1 ? 0x7fffffffu : 0u;  // tests / A-B runs: ranges | waves
        t.choice_waves = plan.tri_blocks * 4u;
        // ranges of the triangle stream: one per wave of the grid while that keeps them short, else ranges of tri_ticket_slots.
        // Their number is bounded by the index buffer (a command that does not fit is reported, not walked past it), by what
        // 32-bit firstIndex can number, and by the largest command times the instances.
        unsigned long long tris = out->culled_index_capacity / 3ull;
        if (tris > 0xffffffffull / 3ull) tris = 0xffffffffull / 3ull;
        // (a command owns floor(indexCount / 3) slots, but the slots are numbered by the running sum of indexCount / 3: index counts that
        //  are no multiple of 3 push later commands up to 2/3 of a slot each — hence the + 1 per instance)
        if (tris > (unsigned long long)n * (ctx->max_lod_tris + 1ull)) tris = (unsigned long long)n * (ctx->max_lod_tris + 1ull);
        size_t need = (size_t)(tris / ctx->tri_ticket_slots) + 2;
        if (need < (size_t)plan.tri_blocks * 4u + 1u) need = (size_t)plan.tri_blocks * 4u + 1u;
        if (sl.chunks_cap < need) {
          if (sl.d_chunk_first) {  // a frame of this slot may still read the old arrays
            MIP_HIP(ctx, hipStreamSynchronize(stream));
            MIP_HIP(ctx, hipFree(sl.d_chunk_first));
            MIP_HIP(ctx, hipFree(sl.d_chunk_status));
            sl.d_chunk_first = nullptr; sl.d_chunk_status = nullptr; sl.chunks_cap = 0;
          }
          MIP_HIP(ctx, hipMalloc(&sl.d_chunk_first, need * 4));
          MIP_HIP(ctx, hipMalloc(&sl.d_chunk_status, need * 8));
          MIP_HIP(ctx, hipMemsetAsync(sl.d_chunk_status, 0, need * 8, stream));
          sl.chunks_cap = need;
          sl.chunk_epoch = 0;
        }
        if (sl.chunk_epoch == 0xffffffffu) {  // tag wrap: start over on a cleared array
          MIP_HIP(ctx, hipMemsetAsync(sl.d_chunk_status, 0, sl.chunks_cap * 8, stream));
          sl.chunk_epoch = 0;
        }
        if (either) {
          // the commands by descending size class for the wave-per-command decomposition; histogram copies, cursors and the ticket live in one block
          if (!sl.d_tri_order) MIP_HIP(ctx, hipMalloc(&sl.d_tri_order, cap * 4));
          if (!sl.d_tri_sort) {
            MIP_HIP(ctx, hipMalloc(&sl.d_tri_sort, mip::kSortWords * 4));
            sl.tri_sort_clean = false;
          }
          // (the re-compaction at the end of a frame clears these counters for the slot's next frame: no clear per frame)
          if (!sl.tri_sort_clean) MIP_HIP(ctx, hipMemsetAsync(sl.d_tri_sort, 0, mip::kSortWords * 4, stream));
          sl.tri_sort_clean = false;  // until this frame's re-compaction has been enqueued
          zero_words = sl.d_tri_sort;
          n_zero = mip::kSortWords;
          t.sort_info = sl.d_tri_sort;
          t.ticket = sl.d_tri_sort + mip::kSortTicket;
          t.final_index_count = sl.d_tmp_final;
        } else {
          if (!sl.tri_ticket_clean) MIP_HIP(ctx, hipMemsetAsync(t.ticket, 0, 4, stream));
          sl.tri_ticket_clean = false;
          zero_words = t.ticket;
          n_zero = 1;
        }
        mip::TriangleChunkArgs ca{};
        ca.t = t;
        ca.t.final_index_count = sl.d_tmp_final;
        ca.range_first_cmd = sl.d_chunk_first;
        ca.range_status = sl.d_chunk_status;
        ca.ranges_cap = (uint32_t)sl.chunks_cap;
        ca.n_waves = plan.tri_blocks * 4u;
        ca.ticket_slots = ctx->tri_ticket_slots;
        ca.epoch = ++sl.chunk_epoch;
        ca.first_index_base = frame->first_index_base;
#ifdef MIP_DEBUG_STAMPS
        if (const char* env = std::getenv("MIP_DEBUG_TILE_ORDER")) ca.debug_reverse = std::strcmp(env, "reverse") == 0 ? 1u : 0u;
        if (const char* env = std::getenv("MIP_DEBUG_SKIP_PART")) ca.debug_skip_part = (uint32_t)std::atoi(env) % 16u + 1u;
#endif
        if (either) {
          ca.t.order = sl.d_tri_order;  // (the wave-per-command decomposition's; the range decomposition does not look at it)
          mip::launch_triangle_stage(plan.tri_map_blocks, plan.tri_blocks, stream, ca);
        } else {
          mip::launch_triangle_cull_chunks(plan.tri_map_blocks, plan.tri_blocks, stream, ca);
        }
      } else if (plan.tri == mip::TriangleKernel::parts) {
        const size_t cap_cmds = ctx->max_instances < ctx->tri_parts_max ? cap : ctx->tri_parts_max;
        if (!sl.d_part_status) {
          MIP_HIP(ctx, hipMalloc(&sl.d_part_status, cap_cmds * mip::kTriParts * 8));
          MIP_HIP(ctx, hipMemsetAsync(sl.d_part_status, 0, cap_cmds * mip::kTriParts * 8, stream));
          sl.tri_epoch = 0;
        }
        if (sl.tri_epoch == 0xffffffffu) {  // tag wrap: start over on a cleared array
          MIP_HIP(ctx, hipMemsetAsync(sl.d_part_status, 0, cap_cmds * mip::kTriParts * 8, stream));
          sl.tri_epoch = 0;
        }
        mip::TrianglePartsArgs pa{};
        pa.t = t;
        pa.t.final_index_count = sl.d_tmp_final;
        pa.part_status = sl.d_part_status;
        pa.epoch = ++sl.tri_epoch;
#ifdef MIP_DEBUG_STAMPS
        if (const char* env = std::getenv("MIP_DEBUG_TILE_ORDER")) pa.debug_reverse = std::strcmp(env, "reverse") == 0 ? 1u : 0u;
        if (const char* env = std::getenv("MIP_DEBUG_SKIP_PART")) pa.debug_skip_part = (uint32_t)std::atoi(env) % mip::kTriParts + 1u;
#endif
        mip::launch_triangle_cull_parts(plan.tri_blocks, stream, pa);
      } else if (plan.tri == mip::TriangleKernel::block) {
        if (plan.tri_reset_ticket) MIP_HIP(ctx, hipMemsetAsync(t.ticket, 0, 4, stream));
        t.pull_tickets = plan.tri_block_tickets ? ctx->tri_batch_from : 0u;
        mip::launch_triangle_cull_block(plan.tri_threads, plan.tri_blocks, stream, t);
      } else {
        if (plan.tri_reset_ticket) MIP_HIP(ctx, hipMemsetAsync(t.ticket, 0, 4, stream));  // only the wave-per-command kernel hands out tickets
        if (plan.tri_either_blocks) {  // both grids; one returns at once (tri_choice_is_block)
          t.index_total = a.index_total;
          t.max_lod_tris = ctx->max_lod_tris;
          if (ctx->tri_force_choice) t.max_lod_tris = ctx->tri_force_choice == 1 ? 0x7fffffffu : 0u;  // tests / A-B runs
          t.pull_tickets = ctx->tri_batch_from;  // (the wave-per-command kernel ignores it: it always pulls single commands)
          mip::launch_triangle_cull_block(256, plan.tri_either_blocks, stream, t);
        }
        mip::launch_triangle_cull_waves(plan.tri_blocks, stream, t);
      }
      MIP_HIP(ctx, hipGetLastError());
      // (re-compacting inside the workgroup kernels, by the last workgroup to finish, was measured: the
      // agent-scope fences it needs cost more than the launch they save — 1 k instances 65 vs 49 us)
      // (the parts and the chunk kernel leave a command's final indexCount beside it: their work items still need the original)
      const bool final_beside = plan.tri == mip::TriangleKernel::parts || plan.tri == mip::TriangleKernel::chunks || plan.tri == mip::TriangleKernel::sorted;
      if (plan.recompact == mip::Recompact::single) {
        mip::RecompactArgs r{};
        r.in_cmds = sl.d_tmp_cmds;
        r.index_count = final_beside ? sl.d_tmp_final : nullptr;
        r.in_count = sl.d_scalars + 2;
        r.out_cmds = (uint32_t*)out->draw_cmds;
        r.out_count = out->draw_count;
        r.zero_words = zero_words;
        r.n_zero = n_zero;
        mip::launch_recompact(stream, r);
      } else {
        mip::RecompactWideArgs r{};
        r.in_cmds = sl.d_tmp_cmds;
        r.index_count = final_beside ? sl.d_tmp_final : nullptr;
        r.in_count = sl.d_scalars + 2;
        r.out_cmds = (uint32_t*)out->draw_cmds;
        r.out_count = out->draw_count;
        r.block_base = (uint32_t*)sl.d_tmp_blocks;
        r.n_blocks = plan.recompact_blocks;
        if (!ctx->tri_recompact_three_launches) {  // one launch (round 5); MIP_TUNE_TRI_RECOMPACT_LAUNCHES=3: round 4's count / scan / scatter
          if (sl.recompact_epoch == 0xffffffffu) {  // tag wrap: start over on cleared granules
            MIP_HIP(ctx, hipMemsetAsync(sl.d_tmp_blocks, 0, (cap / 1024 + 1) * 8, stream));
            sl.recompact_epoch = 0;
          }
          r.block_status = sl.d_tmp_blocks;
          r.epoch = ++sl.recompact_epoch;
          r.help_counter = ctx->d_help + mip::kHelpShards;
          r.zero_words = zero_words;
          r.n_zero = n_zero;
#ifdef MIP_DEBUG_STAMPS
          r.debug_skip = std::getenv("MIP_DEBUG_SKIP_PART") ? 1u : 0u;
#endif
        }
        mip::launch_recompact_wide(stream, r);
      }
      MIP_HIP(ctx, hipGetLastError());
      // the counters are cleared by the re-compaction that has just been enqueued (not by the three-launch form)
      const bool cleared = n_zero != 0u && (plan.recompact == mip::Recompact::single || !ctx->tri_recompact_three_launches);
      if (zero_words == sl.d_tri_sort && zero_words) sl.tri_sort_clean = cleared;
      else if (zero_words) sl.tri_ticket_clean = cleared;
    }
    if (timing) MIP_HIP(ctx, hipEventRecord(ctx->ev1, stream));
  }

  if (!device_out) {
    if (n) {
      if (out->model) MIP_HIP(ctx, hipMemcpyAsync(out->model, ctx->s_model, (size_t)n * 64, hipMemcpyDeviceToHost, stream));
      if (out->visible_bitmap) MIP_HIP(ctx, hipMemcpyAsync(out->visible_bitmap, ctx->s_bitmap, (size_t)words * 4, hipMemcpyDeviceToHost, stream));
      if (out->world_aabb) MIP_HIP(ctx, hipMemcpyAsync(out->world_aabb, ctx->s_aabb, (size_t)n * 24, hipMemcpyDeviceToHost, stream));
    }
    uint32_t scalars[2] = {0, 0};
    if (out->draw_cmds) {
      MIP_HIP(ctx, hipMemcpyAsync(scalars, sl.d_scalars, 8, hipMemcpyDeviceToHost, stream));
      MIP_HIP(ctx, hipStreamSynchronize(stream));
      if (scalars[0] > n) return fail(ctx, MIP_ERR_DEVICE, "draw_count %u > n %u", scalars[0], n);
      if (scalars[0])
        MIP_HIP(ctx, hipMemcpyAsync(out->draw_cmds, ctx->s_cmds, (size_t)scalars[0] * 20, hipMemcpyDeviceToHost, stream));
      *out->draw_count = scalars[0];
      if (out->draw_index_total) *out->draw_index_total = scalars[1];
    }
    MIP_HIP(ctx, hipStreamSynchronize(stream));
  } else if (!async) {
    MIP_HIP(ctx, hipStreamSynchronize(stream));
  }

  if (async) {
    ctx->pending_async = true;
    return MIP_OK;
  }
  if (timing && n) {
    float ms = 0.f;
    MIP_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->timings.runs += 1;
    ctx->timings.last_kernel_ms = ms;
    ctx->timings.total_kernel_ms += ms;
  }
  return check_device_error(ctx);
}

namespace {

// mip_run_many with the launches recorded once and replayed: per slot a linear hipGraph of
// G launches of the instance kernel. Two things change from launch to launch:
//  - the prefix tag: a chain bakes the tags base+1 .. base+G. Replaying the same tags is sound because
//    every launch rewrites every level-0 word and group start it later reads, so the only stale tag a
//    word can hold is the previous launch's — base+G before the chain's first launch (G >= 2) — and
//    because G is even, so the accumulator buffer the first launch adds into is the one the last
//    launch zeroed;
//  - the frame (camera planes, LOD reference point, bases): NOT baked. Node j of a chain reads entry j
//    of the slot's frame ring in device memory (KernelArgs.frame_ring); the host refreshes the ring
//    with one stream-ordered copy in front of every replay. A renderer moves its camera every frame
//    (project_camera runs in the frame loop, src/ecs.rs:66-91, src/main.rs:907-926): the recorded
//    graphs survive that, and only a change of outputs, instance count or kernel choice re-records.
static void frame_words(const MipFrame& f, uint32_t* w) {
  std::memcpy(w, f.planes, 24 * 4);
  std::memcpy(w + 24, f.cam_pos, 3 * 4);
  w[27] = f.first_instance_base;
  w[28] = f.first_index_base;
  w[29] = w[30] = w[31] = 0;
}

static void destroy_graph_set(MipContext::GraphSet& gs) {
  for (auto& fg : gs.per_slot) {
    if (fg.exec) (void)hipGraphExecDestroy(fg.exec);
    if (fg.graph) (void)hipGraphDestroy(fg.graph);
  }
  gs.per_slot.clear();
}

static int32_t run_many_graphed(MipContext* ctx, const MipFrame* frames, uint32_t n_frames, uint32_t first_step,
                                const MipOutputs* outputs, uint32_t n_outputs, uint32_t rounds, uint32_t frames_per_slot) {
  const uint32_t F = (uint32_t)ctx->slots.size();
  const uint32_t G = frames_per_slot;
  for (auto& sl : ctx->slots) {
    if (int32_t rc = reset_prefix_state_if_needed(ctx, sl, G + 2)) return rc;
    if (sl.frame_ring_frames < G) {  // ring + two pinned staging halves, sized for one chain
      MIP_HIP(ctx, hipStreamSynchronize(sl.stream));
      (void)hipFree(sl.d_frame_ring);
      if (sl.h_frame_stage) (void)hipHostFree(sl.h_frame_stage);
      sl.d_frame_ring = nullptr;
      sl.h_frame_stage = nullptr;
      sl.frame_ring_frames = 0;
      MIP_HIP(ctx, hipMalloc(&sl.d_frame_ring, (size_t)G * mip::kFrameWords * 4));
      MIP_HIP(ctx, hipHostMalloc(&sl.h_frame_stage, (size_t)2 * G * mip::kFrameWords * 4, hipHostMallocDefault));
      for (auto& e : sl.stage_free)
        if (!e) MIP_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      sl.frame_ring_frames = G;
      sl.stage_next = 0;
      ctx->graph_generation++;  // recorded nodes point into the old ring
    }
  }

  MipContext::GraphSet* set = nullptr;
  for (size_t i = 0; i < ctx->graph_sets.size();) {
    auto& gs = ctx->graph_sets[i];
    if (gs.generation != ctx->graph_generation) {  // recorded against another instance count, kernel or a cleared state
      destroy_graph_set(gs);
      ctx->graph_sets.erase(ctx->graph_sets.begin() + (long)i);
      continue;
    }
    if (gs.first_slot == ctx->next_slot && gs.frames_per_slot == G && gs.outs.size() == n_outputs &&
        std::memcmp(gs.outs.data(), outputs, sizeof(MipOutputs) * n_outputs) == 0)
      set = &gs;
    ++i;
  }
  if (set)
    for (uint32_t i = 0; i < F; ++i)
      if (ctx->slots[(set->first_slot + i) % F].last_tag == set->per_slot[i].base_epoch + 1) set = nullptr;  // cannot happen; re-record if it does
  if (!set) {
    if (ctx->graph_sets.size() >= 4) {
      destroy_graph_set(ctx->graph_sets.front());
      ctx->graph_sets.erase(ctx->graph_sets.begin());
    }
    // built aside and moved into the cache only when every slot's chain has instantiated: a failure
    // half-way must not leave an entry with null graphs that a later call would match and launch
    MipContext::GraphSet gs;
    gs.outs.assign(outputs, outputs + n_outputs);
    gs.first_slot = ctx->next_slot;
    gs.frames_per_slot = G;
    gs.generation = ctx->graph_generation;
    gs.per_slot.resize(F);
    const int32_t rc = [&]() -> int32_t {
      MipFrame blank{};
      for (uint32_t i = 0; i < F; ++i) {
        MipContext::FrameSlot& sl = ctx->slots[(gs.first_slot + i) % F];
        MipContext::FrameGraph& fg = gs.per_slot[i];
        uint32_t base = sl.epoch > sl.last_tag ? sl.epoch : sl.last_tag;
        if (sl.zero_buf != 2 && ((base + 1) & 1u) != sl.zero_buf) ++base;
        fg.base_epoch = base;
        MIP_HIP(ctx, hipGraphCreate(&fg.graph, 0));
        hipGraphNode_t prev = nullptr;
        for (uint32_t j = 0; j < G; ++j) {
          const MipOutputs* out = &outputs[(i + j * F) % n_outputs];
          mip::KernelArgs a{};
          fill_kernel_args(ctx, sl, &blank, out, true, a);
          a.frame_ring = sl.d_frame_ring + (size_t)j * mip::kFrameWords;
          a.epoch = base + 1 + j;
          mip::FrameKernelParams params(a);
          hipKernelNodeParams kp{};
          const mip::LaunchPlan plan = mip::plan_frame(plan_state(ctx), plan_request(out, false));  // validated by mip_run_many
          a.group_shift = plan.group_shift;
          kp.func = (void*)frame_kernel_of(plan, a.first_mover_rule == 1u);
          kp.gridDim = dim3(plan.n_tiles);
          kp.blockDim = dim3(mip::kTile);
          kp.sharedMemBytes = ctx->lds_pad;
          kp.kernelParams = params.p;
          kp.extra = nullptr;
          hipGraphNode_t node = nullptr;
          MIP_HIP(ctx, hipGraphAddKernelNode(&node, fg.graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
          prev = node;
        }
        MIP_HIP(ctx, hipGraphInstantiate(&fg.exec, fg.graph, nullptr, nullptr, 0));
      }
      return MIP_OK;
    }();
    if (rc != MIP_OK) {
      destroy_graph_set(gs);
      return rc;
    }
    ctx->graph_sets.push_back(std::move(gs));
    ctx->timings.graph_records += 1;
    set = &ctx->graph_sets.back();
  }

  for (uint32_t r = 0; r < rounds; ++r)
    for (uint32_t i = 0; i < F; ++i) {
      MipContext::FrameSlot& sl = ctx->slots[(set->first_slot + i) % F];
      const MipContext::FrameGraph& fg = set->per_slot[i];
      const uint32_t first_buf = (fg.base_epoch + 1) & 1u;
      if (sl.zero_buf != 2 && sl.zero_buf != first_buf)  // other launches ran in between: zero the buffer the chain starts in
        MIP_HIP(ctx, hipMemsetAsync(sl.d_status + ctx->acc1_offset_words + (size_t)first_buf * ctx->groups_cap * mip::kAccStrideWords, 0,
                                    (size_t)ctx->groups_cap * mip::kAccStrideWords * 8, sl.stream));
      // this replay's frames -> a free staging half -> the ring (stream-ordered behind the previous replay)
      const uint32_t half = sl.stage_next;
      sl.stage_next ^= 1u;
      MIP_HIP(ctx, hipEventSynchronize(sl.stage_free[half]));  // the copy that last read this half has finished (no-op if never recorded)
      uint32_t* stage = sl.h_frame_stage + (size_t)half * G * mip::kFrameWords;
      for (uint32_t j = 0; j < G; ++j) {
        const uint64_t step = (uint64_t)first_step + (uint64_t)r * G * F + i + (uint64_t)j * F;
        frame_words(frames[step % n_frames], stage + (size_t)j * mip::kFrameWords);
      }
      MIP_HIP(ctx, hipMemcpyAsync(sl.d_frame_ring, stage, (size_t)G * mip::kFrameWords * 4, hipMemcpyHostToDevice, sl.stream));
      MIP_HIP(ctx, hipEventRecord(sl.stage_free[half], sl.stream));
      MIP_HIP(ctx, hipGraphLaunch(fg.exec, sl.stream));
      sl.last_tag = fg.base_epoch + G;
      if (sl.epoch < sl.last_tag) sl.epoch = sl.last_tag;
      sl.zero_buf = first_buf;  // G is even: the last launch zeroed the buffer the first one uses
      ctx->timings.graph_frames += G;
    }
  ctx->pending_async = true;
  return MIP_OK;
}

}  // namespace
}  // namespace mip_host

using namespace mip_host;

extern "C" {

int32_t mip_run_many(MipContext* ctx, const MipFrame* frames, uint32_t n_frames, const MipOutputs* outputs, uint32_t n_outputs,
                     uint32_t steps) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!frames || n_frames == 0 || !outputs || n_outputs == 0)
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "frames/outputs is NULL or empty");
  bool plain = true;  // only launches of the instance kernel alone are recorded
  for (uint32_t k = 0; k < n_outputs; ++k) {
    if ((outputs[k].flags & (MIP_OUT_DEVICE | MIP_OUT_ASYNC)) != (MIP_OUT_DEVICE | MIP_OUT_ASYNC))
      return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "mip_run_many needs MIP_OUT_DEVICE | MIP_OUT_ASYNC outputs");
    if (outputs[k].culled_index_buffer || !outputs[k].draw_cmds) plain = false;
  }
  uint32_t done = 0;
  const uint32_t F = (uint32_t)ctx->slots.size();
  if (plain && ctx->graph_round && ctx->n && !(ctx->cfg_flags & MIP_CFG_TIMING)) {
    // a round = the smallest run after which slot and output rotation repeat, with an even
    // number of frames per slot, scaled up to about graph_round frames
    uint32_t a = F, b = n_outputs;
    while (b) { const uint32_t t = a % b; a = b; b = t; }
    const uint64_t unit = 2ull * F / a * n_outputs;
    if (unit <= ctx->graph_round && steps >= unit) {
      const uint32_t round = (uint32_t)(ctx->graph_round / unit * unit);
      const uint32_t rounds = steps / round;
      if (rounds) {
        for (uint32_t k = 0; k < n_outputs; ++k)
          {
            mip::LaunchPlan plan;
            if (int32_t rc = plan_for(ctx, &frames[0], &outputs[k], false, &plan)) return rc;
          }
        if (int32_t rc = bind_device(ctx)) return rc;
        if (int32_t rc = run_many_graphed(ctx, frames, n_frames, 0, outputs, n_outputs, rounds, round / F)) return rc;
        done = rounds * round;
      }
    }
  }
  // the rest (or everything) one launch at a time; round % n_outputs == 0 keeps the rotation
  for (uint32_t k = done; k < steps; ++k)
    if (int32_t rc = mip_run(ctx, &frames[k % n_frames], &outputs[k % n_outputs])) return rc;
  return MIP_OK;
}

int32_t mip_run(MipContext* ctx, const MipFrame* frame, const MipOutputs* out) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  return run_frame(ctx, frame, out, false, nullptr);
}

static int32_t run_views_chunk(MipContext* ctx, const MipFrame* frames, const MipOutputs* outs, uint32_t n_views, bool async) {
  if (!frames || !outs) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "frames/outs is NULL");
  if (!ctx->have_instances || !ctx->have_meshes) return fail(ctx, MIP_ERR_NOT_READY, "instances or mesh table not set");
  for (uint32_t v = 0; v < n_views; ++v) {
    const MipOutputs& o = outs[v];
    if (!(o.flags & MIP_OUT_DEVICE)) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "view %u: mip_run_views needs MIP_OUT_DEVICE outputs", v);
    if (!o.draw_cmds || !o.draw_count) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "view %u: draw_cmds and draw_count are required", v);
    if (o.model || o.world_aabb || o.tlas_instances || o.culled_index_buffer)
      return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "view %u: only visible_bitmap, draw_cmds, draw_count and draw_index_total are per view", v);
  }
  if (int32_t rc = bind_device(ctx)) return rc;
  const uint32_t n = ctx->n;
  hipStream_t stream = ctx->stream;
  if (n == 0) {
    for (uint32_t v = 0; v < n_views; ++v) {
      MIP_HIP(ctx, hipMemsetAsync(outs[v].draw_count, 0, 4, stream));
      if (outs[v].draw_index_total) MIP_HIP(ctx, hipMemsetAsync(outs[v].draw_index_total, 0, 4, stream));
    }
  } else {
    if (ctx->view_states.empty()) {
      // built aside: a failed allocation must not leave entries with a null prefix state behind
      std::vector<MipContext::FrameSlot> states(mip::kMaxViews);
      const int32_t rc = [&]() -> int32_t {
        for (auto& vs : states) {
          vs.stream = stream;  // not owned
          MIP_HIP(ctx, hipMalloc(&vs.d_status, ctx->status_bytes));
          MIP_HIP(ctx, hipMemsetAsync(vs.d_status, 0, ctx->status_bytes, stream));
        }
        return MIP_OK;
      }();
      if (rc != MIP_OK) {
        for (auto& vs : states) (void)hipFree(vs.d_status);
        return rc;
      }
      ctx->view_states = std::move(states);
    }
    mip::ViewsArgs a{};
    a.pos = ctx->d_pos; a.rot = ctx->d_rot; a.scale = ctx->d_scale; a.mesh_id = ctx->d_mesh_id;
    a.meshes = ctx->d_meshes; a.mesh_draw = ctx->d_mesh_draw;
    a.n = n;
    a.n_tiles = tiles_for(n);
    a.bitmap_words = (n + 31u) / 32u;
    a.n_views = n_views;
#ifdef MIP_DEBUG_STAMPS
    if (const char* env = std::getenv("MIP_DEBUG_TILE_ORDER"))
      if (a.n_tiles > 1u && std::strcmp(env, "reverse") == 0) a.debug_tile_mult = a.debug_tile_add = a.n_tiles - 1u;
#endif
    for (uint32_t v = 0; v < n_views; ++v) {
      MipContext::FrameSlot& vs = ctx->view_states[v];
      if (int32_t rc = reset_prefix_state_if_needed(ctx, vs, 2)) return rc;
      uint32_t e = vs.epoch + 1;
      if (vs.zero_buf != 2 && (e & 1u) != vs.zero_buf) ++e;
      vs.epoch = vs.last_tag = e;
      vs.zero_buf = (e & 1u) ^ 1u;
      mip::ViewArgs& w = a.view[v];
      w.status0 = vs.d_status;
      w.acc1 = vs.d_status + ctx->acc1_offset_words;
      w.start1 = vs.d_status + ctx->start1_offset_words;
      w.groups_cap = ctx->groups_cap;
      w.group_shift = a.n_tiles <= 512 ? 4u : (a.n_tiles <= 2048 ? 5u : 6u);
      w.epoch = e;
      w.error_flag = ctx->d_error;
      w.help_counter = ctx->d_help + mip::kHelpShards;
      w.bitmap = outs[v].visible_bitmap;
      w.cmds = (uint32_t*)outs[v].draw_cmds;
      w.draw_count = outs[v].draw_count;
      w.index_total = outs[v].draw_index_total;
      w.first_instance_base = frames[v].first_instance_base;
      w.first_index_base = frames[v].first_index_base;
      std::memcpy(w.planes, frames[v].planes, sizeof w.planes);
      std::memcpy(w.cam, frames[v].cam_pos, sizeof w.cam);
    }
    const bool general = ctx->nonfinite_instances != 0 || ctx->force_general;
    if (general) ctx->timings.general_launches += 1;
    mip::launch_cull_views(general, a.n_tiles, stream, a);
    MIP_HIP(ctx, hipGetLastError());
  }
  if (async) {
    ctx->pending_async = true;
    return MIP_OK;
  }
  MIP_HIP(ctx, hipStreamSynchronize(stream));
  return check_device_error(ctx);
}

int32_t mip_run_views(MipContext* ctx, const MipFrame* frames, const MipOutputs* outs, uint32_t n_views) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!frames || !outs) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "frames/outs is NULL");
  if (n_views == 0 || n_views > MIP_MAX_VIEWS) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "n_views %u outside 1..%u", n_views, (unsigned)MIP_MAX_VIEWS);
  const bool async = (outs[0].flags & MIP_OUT_ASYNC) != 0;
  // four views per launch (one wave of a workgroup finishes one view); more views are more launches on the same stream
  for (uint32_t first = 0; first < n_views; first += mip::kMaxViews) {
    const uint32_t k = n_views - first < mip::kMaxViews ? n_views - first : mip::kMaxViews;
    const bool last = first + k == n_views;
    if (int32_t rc = run_views_chunk(ctx, frames + first, outs + first, k, async || !last)) return rc;
  }
  return MIP_OK;
}
int32_t mip_run_skinned(MipContext* ctx, const MipFrame* frame, const MipOutputs* out, void* palette) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!out) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "frame/out is NULL");
  if (!(out->flags & MIP_OUT_DEVICE)) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "mip_run_skinned needs MIP_OUT_DEVICE outputs");
  if (out->culled_index_buffer) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "the per-triangle stage does not skin vertices");
  if (!ctx->n_joints || ctx->poses_n != ctx->n || (ctx->n && !ctx->d_poses))
    return fail(ctx, MIP_ERR_NOT_READY, "skeleton or poses not set for the resident instances");
  if (int32_t rc = bind_device(ctx)) return rc;
  return run_frame(ctx, frame, out, true, palette);
}

int32_t mip_light_draw_lists(MipContext* ctx, const float* light_pos_xyz, uint32_t n_lights, uint32_t first_instance_base,
                             void* out_cmds, int32_t async) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!ctx->have_instances || !ctx->have_meshes) return fail(ctx, MIP_ERR_NOT_READY, "instances or mesh table not set");
  if (!light_pos_xyz || !out_cmds) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "NULL pointer");
  static_assert(MIP_MAX_LIGHTS == mip::kMaxLights, "light limit");
  if (n_lights == 0 || n_lights > MIP_MAX_LIGHTS)
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "n_lights %u outside 1..%u", n_lights, (unsigned)MIP_MAX_LIGHTS);
  if (int32_t rc = bind_device(ctx)) return rc;
  if (ctx->n) {
    mip::LightListArgs a{};
    a.pos = ctx->d_pos;
    a.mesh_id = ctx->d_mesh_id;
    a.meshes = ctx->d_meshes;
    a.mesh_draw = ctx->d_mesh_draw;
    a.out = (uint32_t*)out_cmds;
    a.n = ctx->n;
    a.n_lights = n_lights;
    a.first_instance_base = first_instance_base;
    std::memcpy(a.light, light_pos_xyz, (size_t)n_lights * 12);
    const bool aligned = (ctx->n % 4u) == 0 && ((uintptr_t)out_cmds % 16u) == 0;
    mip::launch_light_draw_lists(aligned, tiles_for(ctx->n), ctx->stream, a);
    MIP_HIP(ctx, hipGetLastError());
  }
  if (async) {
    ctx->pending_async = true;
    return MIP_OK;
  }
  MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return check_device_error(ctx);
}

int32_t mip_wait(MipContext* ctx) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (int32_t rc = bind_device(ctx)) return rc;
  if (int32_t rc = sync_all(ctx)) return rc;
  if (int32_t rc = interop_drain(ctx)) return rc;  // the signals of external semaphores behind the drained frames have gone out
  ctx->pending_async = false;
  int32_t rc = check_device_error(ctx);
  ctx->sharded_pending = 0;
  return rc;
}

}  // extern "C"
