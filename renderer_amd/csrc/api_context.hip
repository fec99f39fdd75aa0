// api_context.hip — C ABI of the instance pipeline (include/mi_instance_pipeline.h), part 1 of 4: the context, its
// resident state (mesh table, instance columns, geometry, skeleton, poses) and the diagnostics. HIP runtime only: no
// torch types, no CPU fallback. Modelled on the reference's one FFI precedent, the vma crate (vma/src/lib.rs:31-64;
// status-code returns as in src/renderer/device/alloc.rs:192-226).
#include "context.hpp"

static_assert(sizeof(MipDrawIndexedIndirectCommand) == 20, "VkDrawIndexedIndirectCommand is 20 bytes");
static_assert(sizeof(MipMesh) == 80, "MipMesh layout");
static_assert(sizeof(MipShardHeader) == 32, "MipShardHeader layout");
static_assert(sizeof(mip::MeshEntry) == 32, "MeshEntry layout");
static_assert(sizeof(mip::KernelArgs) <= 4096, "kernel argument block");

namespace mip_host {

int32_t fail(MipContext* ctx, int32_t code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
    va_end(ap);
  }
  return code;
}

int32_t bind_device(MipContext* ctx) {
  MIP_HIP(ctx, hipSetDevice(ctx->device));
  return MIP_OK;
}

int32_t sync_all(MipContext* ctx) {
  for (auto& sl : ctx->slots) MIP_HIP(ctx, hipStreamSynchronize(sl.stream));
  return MIP_OK;
}

// Reads and clears the device-visible error words (one per kind, instance_kernel.hpp) after the calling entry point has
// drained the stream(s) it is responsible for.
//
// A sharded frame may need a COLLECTIVE repair (the all-gather + merge repeated at full capacity when a tightened
// chunk overflowed). Whether it does is decided from the gathered headers alone — kErrChunkOverflow is raised by the
// merge kernel, which sees the same headers on every rank — and never from anything only this rank knows.
//
// A SYNCHRONOUS call made while asynchronous work of the same context is still in flight (frames on other slots) has
// drained only its own stream: an error word it finds may belong to one of those frames. It is reported at once AND kept
// (carried_error_bits) for the mip_wait that ends them, so that wait cannot return MIP_OK for a frame whose outputs are
// invalid (round-3 advisor finding).
int32_t check_device_error(MipContext* ctx) {
  uint32_t e = 0;
  for (uint32_t k = 0; k < mip::kErrWords; ++k) {
    e |= ((volatile uint32_t*)ctx->h_error)[k];
    ((volatile uint32_t*)ctx->h_error)[k] = 0;
  }
  if (ctx->pending_async) ctx->carried_error_bits |= e & ~mip::kErrChunkOverflow;  // (an overflow is repaired below, once)
  else { e |= ctx->carried_error_bits; ctx->carried_error_bits = 0; }
  ctx->last_error_bits = e;
  if (!e) return MIP_OK;
  int32_t repair_rc = MIP_OK;
  if (e & mip::kErrChunkOverflow) repair_rc = repair_sharded_overflow(ctx);
  if (e & mip::kErrIndexOverflow)
    return fail(ctx, MIP_ERR_CAPACITY, "culled_index_buffer too small for a command's index range; its triangles were dropped");
  if (e & mip::kErrSemaphore)
    return fail(ctx, MIP_ERR_TIMEOUT, "a wait for (or signal of) an external semaphore failed or expired after 10 s; the frame behind it ran anyway");
  if (e & mip::kErrWireRecord)
    return fail(ctx, MIP_ERR_DEVICE, "a wire record names a mesh outside this context's mesh table (corrupt chunk, or the ranks hold different tables)");
  return repair_rc;
}

// Number of instances of [first, first + count) of the resident columns that fail the finite test, and (bad_ids != null)
// how many of them name a mesh outside a table of `m` entries.
// Synchronous (uploads are): one small kernel and an 8-byte read-back on the upload stream.
int32_t census(MipContext* ctx, uint32_t first, uint32_t count, uint32_t* out, uint32_t* bad_ids, uint32_t m) {
  *out = 0;
  if (bad_ids) *bad_ids = 0;
  if (!count) return MIP_OK;
  MIP_HIP(ctx, hipMemsetAsync(ctx->d_census, 0, 8, ctx->stream));
  mip::CensusArgs c{};
  c.pos = ctx->d_pos; c.rot = ctx->d_rot; c.scale = ctx->d_scale;
  c.mesh_id = bad_ids ? ctx->d_mesh_id : nullptr;
  c.n_meshes = m;
  c.first = first; c.count = count; c.out = ctx->d_census;
  c.box_abs = ctx->box_abs;
  uint32_t blocks = (count + 255u) / 256u;
  if (blocks > 2048u) blocks = 2048u;
  hipLaunchKernelGGL(mip::mip_count_nonfinite_kernel, dim3(blocks), dim3(256), 0, ctx->stream, c);
  MIP_HIP(ctx, hipGetLastError());
  uint32_t both[2] = {0, 0};
  MIP_HIP(ctx, hipMemcpyAsync(both, ctx->d_census, 8, hipMemcpyDeviceToHost, ctx->stream));
  MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *out = both[0];
  if (bad_ids) *bad_ids = both[1];
  return MIP_OK;
}

namespace {
void free_all(MipContext* ctx) {
  if (!ctx) return;
  if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
  for (auto& sl : ctx->slots)
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
  drop_graphs(ctx);
  (void)interop_drain(ctx);  // the streams have drained: let queued semaphore signals go out before the helper threads stop
  interop_release(ctx);
  comm_release(ctx);
  (void)hipFree(ctx->d_pos);
  (void)hipFree(ctx->d_rot);
  (void)hipFree(ctx->d_scale);
  (void)hipFree(ctx->d_mesh_id);
  (void)hipFree(ctx->d_meshes);
  (void)hipFree(ctx->d_census);
  (void)hipFree(ctx->d_help);
  (void)hipFree(ctx->d_mesh_draw);
  (void)hipFree(ctx->d_blas);
  (void)hipFree(ctx->d_vertices);
  (void)hipFree(ctx->d_indices);
  (void)hipFree(ctx->d_joints);
  (void)hipFree(ctx->d_poses_owned);
  for (auto& sl : ctx->view_states) (void)hipFree(sl.d_status);
  for (auto& sl : ctx->slots) {
    (void)hipFree(sl.d_status);
    (void)hipFree(sl.d_scalars);
    (void)hipFree(sl.d_tmp_cmds);
    (void)hipFree(sl.d_tmp_src);
    (void)hipFree(sl.d_tmp_blocks);
    (void)hipFree(sl.d_tmp_final);
    (void)hipFree(sl.d_part_status);
    (void)hipFree(sl.d_tri_order);
    (void)hipFree(sl.d_tri_sort);
    (void)hipFree(sl.d_chunk_first);
    (void)hipFree(sl.d_chunk_status);
    (void)hipFree(sl.d_skin_box);
    (void)hipFree(sl.d_frame_ring);
    if (sl.h_frame_stage) (void)hipHostFree(sl.h_frame_stage);
    for (auto& e : sl.stage_free)
      if (e) (void)hipEventDestroy(e);
  }
  (void)hipFree(ctx->s_model);
  (void)hipFree(ctx->s_bitmap);
  (void)hipFree(ctx->s_cmds);
  (void)hipFree(ctx->s_aabb);
#ifdef MIP_DEBUG_STAMPS
  (void)hipFree(ctx->d_stamps);
#endif
  if (ctx->h_error) (void)hipHostFree(ctx->h_error);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  for (auto& sl : ctx->slots)
    if (sl.own_stream && sl.stream) (void)hipStreamDestroy(sl.stream);
  delete ctx;
}
}  // namespace

}  // namespace mip_host

using namespace mip_host;

extern "C" {

uint32_t mip_abi_version(void) { return MIP_ABI_VERSION; }

int32_t mip_create(const MipConfig* cfg, MipContext** out) {
  if (out) *out = nullptr;
  if (!cfg || !out || cfg->struct_size != sizeof(MipConfig)) return MIP_ERR_INVALID_ARGUMENT;
  if (cfg->max_instances > 0x3fffffffu) return MIP_ERR_INVALID_ARGUMENT;
  if (cfg->frames_in_flight > MIP_MAX_FRAMES_IN_FLIGHT) return MIP_ERR_INVALID_ARGUMENT;
  if (cfg->frames_in_flight > 1 && cfg->stream) return MIP_ERR_INVALID_ARGUMENT;  // one caller stream cannot overlap frames
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MIP_ERR_NO_DEVICE;
  if (cfg->device_ordinal < 0 || cfg->device_ordinal >= count) return MIP_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device_ordinal) != hipSuccess) return MIP_ERR_NO_DEVICE;
  // The code object is built for gfx950 only; anything else could not launch it.
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MIP_ERR_NO_DEVICE;

  MipContext* ctx = new (std::nothrow) MipContext();
  if (!ctx) return MIP_ERR_OUT_OF_MEMORY;
  ctx->device = cfg->device_ordinal;
  ctx->max_instances = cfg->max_instances;
  ctx->max_meshes = cfg->max_meshes;
  ctx->cfg_flags = cfg->flags;

  int32_t rc = [&]() -> int32_t {
    MIP_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t frames = cfg->frames_in_flight ? cfg->frames_in_flight : 1u;
    ctx->slots.resize(frames);
    for (auto& sl : ctx->slots) {
      if (cfg->stream) {
        sl.stream = (hipStream_t)cfg->stream;  // frames == 1 (checked above)
      } else {
        MIP_HIP(ctx, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        sl.own_stream = true;
      }
    }
    ctx->stream = ctx->slots[0].stream;
    const size_t cap = ctx->max_instances ? ctx->max_instances : 1;
    const size_t mcap = ctx->max_meshes ? ctx->max_meshes : 1;
    MIP_HIP(ctx, hipMalloc(&ctx->d_pos, cap * 12));
    MIP_HIP(ctx, hipMalloc(&ctx->d_rot, cap * 16));
    MIP_HIP(ctx, hipMalloc(&ctx->d_scale, cap * 4));
    MIP_HIP(ctx, hipMalloc(&ctx->d_mesh_id, cap * 4));
    MIP_HIP(ctx, hipMalloc(&ctx->d_meshes, mcap * sizeof(mip::MeshEntry)));
    MIP_HIP(ctx, hipMalloc(&ctx->d_mesh_draw, mcap * sizeof(mip::MeshDraw)));
    MIP_HIP(ctx, hipMalloc(&ctx->d_census, 8));
    ctx->cu_count = prop.multiProcessorCount;
    const size_t tiles_cap = tiles_for((uint32_t)cap);
    // smallest group the kernel may pick is 16 tiles (group_shift 4)
    ctx->groups_cap = (uint32_t)((tiles_cap + 15) / 16);
    ctx->acc1_offset_words = (uint32_t)((tiles_cap + 31) / 32 * 32);  // keep the accumulators 256-B aligned
    ctx->start1_offset_words = ctx->acc1_offset_words + ctx->groups_cap * 2 * mip::kAccStrideWords;
    ctx->helps_seen_offset_words = ctx->start1_offset_words + ctx->groups_cap * 2;  // one more granule: KernelArgs.helps_seen
    ctx->status_bytes = ((size_t)ctx->helps_seen_offset_words + 1) * 8;
    for (auto& sl : ctx->slots) {
      MIP_HIP(ctx, hipMalloc(&sl.d_status, ctx->status_bytes));
      MIP_HIP(ctx, hipMemset(sl.d_status, 0, ctx->status_bytes));  // epoch 0 is never used
      MIP_HIP(ctx, hipMalloc(&sl.d_scalars, 64));
      MIP_HIP(ctx, hipMemset(sl.d_scalars, 0, 64));
    }
    MIP_HIP(ctx, hipHostMalloc(&ctx->h_error, 64, hipHostMallocMapped));
    std::memset(ctx->h_error, 0, 64);
    MIP_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->d_error, ctx->h_error, 0));
    MIP_HIP(ctx, hipMalloc(&ctx->d_help, kHelpWords * 4));  // [0, kHelpShards): the frame kernel's helps, [kHelpShards]: every other kernel's
    MIP_HIP(ctx, hipMemset(ctx->d_help, 0, kHelpWords * 4));
#ifdef MIP_DEBUG_STAMPS
    MIP_HIP(ctx, hipMalloc(&ctx->d_stamps, tiles_cap * 64));
    MIP_HIP(ctx, hipMemset(ctx->d_stamps, 0, tiles_cap * 64));
#endif
    if (const char* env = std::getenv("MIP_TUNE_LDS_PAD")) ctx->lds_pad = (uint32_t)std::atoi(env);
    ctx->no_one_mesh = std::getenv("MIP_TUNE_NO_ONE_MESH") != nullptr;
    if (const char* env = std::getenv("MIP_TUNE_FIRST_MOVER"))  // the frame kernel's first-mover rule: always | never | (default) when the previous launch helped
      ctx->first_mover_env = std::strcmp(env, "always") == 0 ? 1u : (std::strcmp(env, "never") == 0 ? 2u : 0u);
    if (const char* env = std::getenv("MIP_TUNE_TRI_BLOCK_THREADS")) {
      const uint32_t v = (uint32_t)std::atoi(env);
      if (v == 256u || v == 512u || v == 1024u) ctx->tri_block_threads = v;
    }
    if (const char* env = std::getenv("MIP_TUNE_TRI_BLOCK_MAX")) ctx->tri_block_max = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("MIP_TUNE_TRI_PARTS_MAX")) ctx->tri_parts_max = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("MIP_TUNE_TRI_RECOMPACT_LAUNCHES")) ctx->tri_recompact_three_launches = std::atoi(env) == 3;
    if (const char* env = std::getenv("MIP_TUNE_TRI_RANGE_SLOTS")) {
      const uint32_t v = (uint32_t)std::strtoul(env, nullptr, 10) / 64u * 64u;
      if (v >= 256u && v <= 8192u) ctx->tri_ticket_slots = v;
    }
    if (const char* env = std::getenv("MIP_TUNE_TRI_CHUNKS_FROM")) ctx->tri_chunks_from = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("MIP_TUNE_TRI_NO_CHOICE")) ctx->tri_no_choice = std::atoi(env) != 0;
    if (const char* env = std::getenv("MIP_TUNE_TRI_CHOICE")) ctx->tri_force_choice = env[0] == 'b' ? 1 : (env[0] == 'w' ? 2 : 0);
    if (const char* env = std::getenv("MIP_TUNE_TRI_BATCH_FROM")) ctx->tri_batch_from = (uint32_t)std::strtoul(env, nullptr, 10);
    if (ctx->tri_batch_from == 0) ctx->tri_batch_from = 1;
    if (const char* env = std::getenv("MIP_TUNE_ORDER")) {
      const int v = std::atoi(env);
      if (v == 1 || v == 3) ctx->force_order = v;
    }
    if (const char* env = std::getenv("MIP_TUNE_FORCE_GENERAL")) ctx->force_general = std::atoi(env) != 0;
    if (const char* env = std::getenv("MIP_TUNE_GRAPH_ROUND")) ctx->graph_round = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("MIP_TEST_EPOCH_START"))  // tests: start next to the tag wrap
      for (auto& sl : ctx->slots) sl.epoch = (uint32_t)std::strtoul(env, nullptr, 10);
    MIP_HIP(ctx, hipEventCreate(&ctx->ev0));
    MIP_HIP(ctx, hipEventCreate(&ctx->ev1));
    // the memsets above ran on the null stream, which the slots' non-blocking streams do not wait for
    MIP_HIP(ctx, hipDeviceSynchronize());
    return MIP_OK;
  }();
  if (rc != MIP_OK) {
    free_all(ctx);
    return rc;
  }
  *out = ctx;
  return MIP_OK;
}

void mip_destroy(MipContext* ctx) { free_all(ctx); }

int32_t mip_set_mesh_table(MipContext* ctx, const MipMesh* meshes, uint32_t m) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!meshes && m) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "meshes is NULL");
  if (m > ctx->max_meshes) return fail(ctx, MIP_ERR_CAPACITY, "%u meshes > max_meshes %u", m, ctx->max_meshes);
  std::vector<mip::MeshEntry> entries(m);
  std::vector<mip::MeshDraw> draw(m);
  for (uint32_t k = 0; k < m; ++k) {
    const MipMesh& s = meshes[k];
    if (s.n_lods < 1 || s.n_lods > MIP_MAX_LODS)
      return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "mesh %u: n_lods %u outside 1..%u", k, s.n_lods, MIP_MAX_LODS);
    for (int a = 0; a < 3; ++a)
      if (!std::isfinite(s.aabb_min[a]) || !std::isfinite(s.aabb_max[a]))
        return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "mesh %u: non-finite bounds", k);
    mip::MeshEntry& e = entries[k];
    e.min_x = s.aabb_min[0]; e.min_y = s.aabb_min[1]; e.min_z = s.aabb_min[2];
    e.max_x = s.aabb_max[0]; e.max_y = s.aabb_max[1]; e.max_z = s.aabb_max[2];
    e.len0 = s.index_len[0];
    e.len1 = s.n_lods > 1 ? s.index_len[1] : s.index_len[0];
    draw[k].vertex_offset = s.vertex_offset;
    draw[k].src_offset0 = s.index_offset[0];
    draw[k].src_offset1 = s.n_lods > 1 ? s.index_offset[1] : s.index_offset[0];
    draw[k].pad = 0;
  }
  if (int32_t rc = bind_device(ctx)) return rc;
  if (int32_t rc = sync_all(ctx)) return rc;
  if (ctx->have_instances && ctx->have_meshes && m < ctx->m) {
    // a smaller table: the resident mesh ids (validated against the old one) must still be inside it
    uint32_t bad = 0, bad_ids = 0;
    if (int32_t rc = census(ctx, 0, ctx->n, &bad, &bad_ids, m)) return rc;
    if (bad_ids) {
      // a new scene: table first, instances next (the documented order). The old instances cannot run against
      // this table — they are no longer resident; a frame before the next upload fails with MIP_ERR_NOT_READY.
      ctx->have_instances = false;
      ctx->n = 0;
      ctx->nonfinite_instances = 0;
      for (auto& sl : ctx->slots) sl.status_dirty = true;
      for (auto& sl : ctx->view_states) sl.status_dirty = true;
      ctx->graph_generation++;
    }
  }
  if (m) {
    MIP_HIP(ctx, hipMemcpyAsync(ctx->d_meshes, entries.data(), m * sizeof(mip::MeshEntry), hipMemcpyHostToDevice, ctx->stream));
    MIP_HIP(ctx, hipMemcpyAsync(ctx->d_mesh_draw, draw.data(), m * sizeof(mip::MeshDraw), hipMemcpyHostToDevice, ctx->stream));
    MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the sources are locals
  }
  // recorded launches carry what follows from the table's SIZE in their arguments (n_meshes, the wire form's index bits, and whether
  // the one entry of a one-mesh table is read as a scalar: KernelArgs.one_mesh): another size, another recording
  if (m != ctx->m) ctx->graph_generation++;
  ctx->m = m;
  ctx->have_meshes = true;
  ctx->h_meshes.assign(meshes, meshes + m);
  ctx->geometry_checked = 0;
  ctx->max_lod_tris = 0;
  for (uint32_t k = 0; k < m; ++k)
    for (uint32_t l = 0; l < meshes[k].n_lods && l < 2u; ++l)
      if (meshes[k].index_len[l] / 3u > ctx->max_lod_tris) ctx->max_lod_tris = meshes[k].index_len[l] / 3u;
  float box_abs = 0.f;
  for (uint32_t k = 0; k < m; ++k) {
    float sum = 0.f;
    for (int a = 0; a < 3; ++a) sum += std::fabs(meshes[k].aabb_min[a]) + std::fabs(meshes[k].aabb_max[a]);
    if (sum > box_abs) box_abs = sum;
  }
  if (box_abs != ctx->box_abs) {  // the census' overflow bound moved: count the resident instances again
    ctx->box_abs = box_abs;
    if (ctx->have_instances) {
      uint32_t bad = 0;
      if (int32_t rc = census(ctx, 0, ctx->n, &bad)) return rc;
      if ((bad != 0) != (ctx->nonfinite_instances != 0)) ctx->graph_generation++;
      ctx->nonfinite_instances = bad;
    }
  }
  return MIP_OK;
}

int32_t mip_set_blas_addresses(MipContext* ctx, const uint64_t* addresses, uint32_t m) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!addresses && m) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "addresses is NULL");
  if (!ctx->have_meshes || m != ctx->m) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "%u addresses for %u meshes", m, ctx->m);
  if (int32_t rc = bind_device(ctx)) return rc;
  if (int32_t rc = sync_all(ctx)) return rc;
  if (!ctx->d_blas) {
    MIP_HIP(ctx, hipMalloc(&ctx->d_blas, (size_t)(ctx->max_meshes ? ctx->max_meshes : 1) * 8));
    ctx->graph_generation++;  // recorded launches carry the old (null) table pointer
  }
  if (m) MIP_HIP(ctx, hipMemcpyAsync(ctx->d_blas, addresses, (size_t)m * 8, hipMemcpyHostToDevice, ctx->stream));
  MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // uploads are stream-ordered copies: finished before any slot launches again
  return MIP_OK;
}

int32_t mip_set_geometry(MipContext* ctx, const float* vertex_xyz, uint32_t n_vertices, const uint32_t* indices,
                         uint32_t n_indices) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if ((n_vertices && !vertex_xyz) || (n_indices && !indices)) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "NULL geometry");
  if (int32_t rc = bind_device(ctx)) return rc;
  if (int32_t rc = sync_all(ctx)) return rc;
  (void)hipFree(ctx->d_vertices);
  (void)hipFree(ctx->d_indices);
  ctx->d_vertices = nullptr;
  ctx->d_indices = nullptr;
  ctx->have_geometry = false;
  MIP_HIP(ctx, hipMalloc(&ctx->d_vertices, (size_t)(n_vertices ? n_vertices : 1) * 12));
  MIP_HIP(ctx, hipMalloc(&ctx->d_indices, (size_t)(n_indices ? n_indices : 1) * 4));
  if (n_vertices) MIP_HIP(ctx, hipMemcpyAsync(ctx->d_vertices, vertex_xyz, (size_t)n_vertices * 12, hipMemcpyHostToDevice, ctx->stream));
  if (n_indices) MIP_HIP(ctx, hipMemcpyAsync(ctx->d_indices, indices, (size_t)n_indices * 4, hipMemcpyHostToDevice, ctx->stream));
  MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // uploads are stream-ordered copies: finished before any slot launches again
  ctx->n_vertices = n_vertices;
  ctx->n_indices = n_indices;
  bool finite = true;
  for (size_t k = 0; k < (size_t)n_vertices * 3 && finite; ++k) finite = std::isfinite(vertex_xyz[k]);
  ctx->geometry_finite = finite;
  ctx->have_geometry = true;
  ctx->h_indices.assign(indices, indices + n_indices);
  ctx->geometry_checked = 0;
  return MIP_OK;
}

static int32_t set_instances_common(MipContext* ctx, const void* pos, const void* rot, const void* scale,
                                    const void* mesh_id, uint32_t n, hipMemcpyKind kind) {
  if (n > ctx->max_instances)
    return fail(ctx, MIP_ERR_CAPACITY, "%u instances > max_instances %u", n, ctx->max_instances);
  if (n && (!pos || !rot || !scale || !mesh_id)) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "NULL instance column");
  if (int32_t rc = bind_device(ctx)) return rc;
  if (int32_t rc = sync_all(ctx)) return rc;
  if (n) {
    MIP_HIP(ctx, hipMemcpyAsync(ctx->d_pos, pos, (size_t)n * 12, kind, ctx->stream));
    MIP_HIP(ctx, hipMemcpyAsync(ctx->d_rot, rot, (size_t)n * 16, kind, ctx->stream));
    MIP_HIP(ctx, hipMemcpyAsync(ctx->d_scale, scale, (size_t)n * 4, kind, ctx->stream));
    MIP_HIP(ctx, hipMemcpyAsync(ctx->d_mesh_id, mesh_id, (size_t)n * 4, kind, ctx->stream));
    MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // stream-ordered copies: finished before any slot launches again
  }
  uint32_t bad = 0, bad_ids = 0;
  if (int32_t rc = census(ctx, 0, n, &bad, &bad_ids, ctx->m)) return rc;
  if (bad_ids) {
    // the resident columns now hold ids the frame kernel would follow out of the mesh table: nothing is resident
    ctx->have_instances = false;
    ctx->n = 0;
    for (auto& sl : ctx->slots) sl.status_dirty = true;
    for (auto& sl : ctx->view_states) sl.status_dirty = true;
    ctx->graph_generation++;
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "%u instance(s) with a mesh id >= %u meshes; no instances are resident now", bad_ids, ctx->m);
  }
  if ((bad != 0) != (ctx->nonfinite_instances != 0)) ctx->graph_generation++;  // recorded launches name the other kernel
  ctx->nonfinite_instances = bad;
  if (n != ctx->n) {
    for (auto& sl : ctx->slots) sl.status_dirty = true;  // tile/group geometry changes with n
    for (auto& sl : ctx->view_states) sl.status_dirty = true;
    ctx->graph_generation++;                             // and so does every recorded launch
  }
  ctx->n = n;
  ctx->have_instances = true;
  return MIP_OK;
}

int32_t mip_set_instances(MipContext* ctx, const float* pos_xyz, const float* rot_ijkw, const float* scale,
                          const uint32_t* mesh_id, uint32_t n) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!ctx->have_meshes) return fail(ctx, MIP_ERR_NOT_READY, "set the mesh table before the instances");
  if (n && mesh_id)
    for (uint32_t i = 0; i < n; ++i)
      if (mesh_id[i] >= ctx->m)
        return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "instance %u: mesh id %u >= %u meshes", i, mesh_id[i], ctx->m);
  return set_instances_common(ctx, pos_xyz, rot_ijkw, scale, mesh_id, n, hipMemcpyHostToDevice);
}

int32_t mip_update_instances(MipContext* ctx, uint32_t first, uint32_t count, const float* pos_xyz, const float* rot_ijkw,
                             const float* scale, const uint32_t* mesh_id) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!ctx->have_instances) return fail(ctx, MIP_ERR_NOT_READY, "no resident instances to update");
  if ((uint64_t)first + count > ctx->n) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "range [%u, %u) exceeds %u instances", first, first + count, ctx->n);
  if (mesh_id)
    for (uint32_t i = 0; i < count; ++i)
      if (mesh_id[i] >= ctx->m)
        return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "instance %u: mesh id %u >= %u meshes", first + i, mesh_id[i], ctx->m);
  if (int32_t rc = bind_device(ctx)) return rc;
  if (int32_t rc = sync_all(ctx)) return rc;
  if (count) {
    uint32_t bad_before = 0, bad_after = 0;
    if (int32_t rc = census(ctx, first, count, &bad_before)) return rc;
    if (pos_xyz) MIP_HIP(ctx, hipMemcpyAsync(ctx->d_pos + (size_t)first * 3, pos_xyz, (size_t)count * 12, hipMemcpyHostToDevice, ctx->stream));
    if (rot_ijkw) MIP_HIP(ctx, hipMemcpyAsync(ctx->d_rot + first, rot_ijkw, (size_t)count * 16, hipMemcpyHostToDevice, ctx->stream));
    if (scale) MIP_HIP(ctx, hipMemcpyAsync(ctx->d_scale + first, scale, (size_t)count * 4, hipMemcpyHostToDevice, ctx->stream));
    if (mesh_id) MIP_HIP(ctx, hipMemcpyAsync(ctx->d_mesh_id + first, mesh_id, (size_t)count * 4, hipMemcpyHostToDevice, ctx->stream));
    MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (int32_t rc = census(ctx, first, count, &bad_after)) return rc;
    const uint64_t total = ctx->nonfinite_instances - bad_before + bad_after;
    if ((total != 0) != (ctx->nonfinite_instances != 0)) ctx->graph_generation++;
    ctx->nonfinite_instances = total;
  }
  return MIP_OK;
}

int32_t mip_set_instances_device(MipContext* ctx, const void* pos_xyz, const void* rot_ijkw, const void* scale,
                                 const void* mesh_id, uint32_t n) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!ctx->have_meshes) return fail(ctx, MIP_ERR_NOT_READY, "set the mesh table before the instances");
  return set_instances_common(ctx, pos_xyz, rot_ijkw, scale, mesh_id, n, hipMemcpyDeviceToDevice);
}

int32_t mip_set_skeleton(MipContext* ctx, const int32_t* parent, const float* inverse_bind, const float* joint_box,
                         uint32_t n_joints) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!parent || !inverse_bind || !joint_box) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "NULL skeleton array");
  static_assert(MIP_MAX_JOINTS == mip::kMaxJoints && MIP_POSE_FLOATS == mip::kPoseWords, "skinning limits");
  if (n_joints == 0 || n_joints > MIP_MAX_JOINTS)
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "n_joints %u outside 1..%u", n_joints, (unsigned)MIP_MAX_JOINTS);
  std::vector<mip::JointEntry> joints(n_joints);
  std::vector<uint32_t> depth(n_joints);
  uint32_t max_depth = 0;
  for (uint32_t k = 0; k < n_joints; ++k) {
    if (parent[k] >= (int32_t)k || parent[k] < -1)
      return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "joint %u: parent %d must be -1 or an earlier joint", k, parent[k]);
    mip::JointEntry& j = joints[k];
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 3; ++r) j.ibm[c * 3 + r] = inverse_bind[(size_t)k * 16 + c * 4 + r];
    std::memcpy(j.box, joint_box + (size_t)k * 6, sizeof j.box);
    j.parent = parent[k];
    depth[k] = parent[k] < 0 ? 0u : depth[parent[k]] + 1u;
    if (depth[k] > max_depth) max_depth = depth[k];
  }
  // joints in depth order (stable): level d owns sorted entries [level_start[d], level_start[d+1])
  uint8_t level_start[mip::kMaxJoints + 2] = {0};
  uint32_t level_inv[mip::kMaxJoints + 1] = {0};
  uint32_t at = 0;
  for (uint32_t d = 0; d <= max_depth; ++d) {
    level_start[d] = (uint8_t)at;
    for (uint32_t k = 0; k < n_joints; ++k)
      if (depth[k] == d) joints[at++].sorted = k | ((uint32_t)(parent[k] < 0 ? 0 : parent[k]) << 8);
    const uint32_t cnt = at - level_start[d];
    level_inv[d] = (65536u + cnt - 1u) / cnt;
  }
  for (uint32_t d = max_depth + 1; d < mip::kMaxJoints + 2; ++d) level_start[d] = (uint8_t)at;
  if (int32_t rc = bind_device(ctx)) return rc;
  if (int32_t rc = sync_all(ctx)) return rc;
  if (!ctx->d_joints) MIP_HIP(ctx, hipMalloc(&ctx->d_joints, sizeof(mip::JointEntry) * MIP_MAX_JOINTS));
  MIP_HIP(ctx, hipMemcpyAsync(ctx->d_joints, joints.data(), sizeof(mip::JointEntry) * n_joints, hipMemcpyHostToDevice, ctx->stream));
  MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (n_joints != ctx->n_joints) {  // the pose layout depends on the joint count
    ctx->d_poses = nullptr;
    ctx->poses_n = 0;
  }
  ctx->n_joints = n_joints;
  ctx->max_joint_depth = max_depth;
  {
    float box_max = 0.0f;
    bool finite = true;
    for (size_t q = 0; q < (size_t)n_joints * 6; ++q) {
      finite = finite && std::isfinite(joint_box[q]);
      box_max = std::fmax(box_max, std::fabs(joint_box[q]));
    }
    ctx->joint_box_bound = finite ? 3.0f * box_max + 1.0f : INFINITY;
  }
  std::memcpy(ctx->joint_level_start, level_start, sizeof level_start);
  std::memcpy(ctx->joint_level_inv, level_inv, sizeof level_inv);
  return MIP_OK;
}

int32_t mip_set_poses(MipContext* ctx, const void* joint_trs, uint32_t n, int32_t device) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!ctx->n_joints) return fail(ctx, MIP_ERR_NOT_READY, "set the skeleton before the poses");
  if (!ctx->have_instances || n != ctx->n) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "%u poses for %u instances", n, ctx->n);
  if (!joint_trs && n) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "joint_trs is NULL");
  if (device) {
    // borrowed: nothing is copied and nothing in flight is touched — frames already queued keep the
    // pointer they were launched with, so an animation system can alternate two buffers
    ctx->d_poses = (const float*)joint_trs;
  } else {
    if (int32_t rc = bind_device(ctx)) return rc;
    if (int32_t rc = sync_all(ctx)) return rc;
    if (!ctx->d_poses_owned)
      MIP_HIP(ctx, hipMalloc(&ctx->d_poses_owned, (size_t)(ctx->max_instances ? ctx->max_instances : 1) * MIP_MAX_JOINTS * MIP_POSE_FLOATS * 4));
    if (n) {
      MIP_HIP(ctx, hipMemcpyAsync(ctx->d_poses_owned, joint_trs, (size_t)n * ctx->n_joints * MIP_POSE_FLOATS * 4, hipMemcpyHostToDevice, ctx->stream));
      MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->d_poses = ctx->d_poses_owned;
  }
  ctx->poses_n = n;
  return MIP_OK;
}

namespace {
// Σ of the device's help words (wrapping)
int32_t read_help_words(MipContext* ctx, uint32_t* out) {
  uint32_t words[kHelpWords];
  MIP_HIP(ctx, hipMemcpy(words, ctx->d_help, sizeof words, hipMemcpyDeviceToHost));
  uint32_t sum = 0;
  for (uint32_t w : words) sum += w;
  *out = sum;
  return MIP_OK;
}
}  // namespace

const char* mip_last_error(const MipContext* ctx) { return ctx ? ctx->err : "null context"; }

int32_t mip_get_timings(MipContext* ctx, MipTimings* out) {
  if (!ctx || !out) return MIP_ERR_INVALID_ARGUMENT;
  if (int32_t rc = bind_device(ctx)) return rc;
  uint32_t helps = 0;  // the device adds (kernels that are in flight may still be adding), the host only reads: a small blocking copy
  if (int32_t rc = read_help_words(ctx, &helps)) return rc;
  ctx->timings.prefix_helps = helps - ctx->help_base;  // (wrapping difference: cumulative since create or mip_reset_timings)
  *out = ctx->timings;
  return MIP_OK;
}

int32_t mip_reset_timings(MipContext* ctx) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (int32_t rc = bind_device(ctx)) return rc;
  // the help counter is written by kernels that may be in flight: the device word is never cleared, the host keeps the value it
  // had at the reset (round 4 cleared it only when nothing was in flight, and reported the old count beside freshly reset fields otherwise)
  uint32_t helps = 0;
  if (int32_t rc = read_help_words(ctx, &helps)) return rc;
  ctx->help_base = helps;
  ctx->timings = MipTimings{};
  return MIP_OK;
}

uint32_t mip_instance_count(const MipContext* ctx) { return ctx ? ctx->n : 0; }

#ifdef MIP_DEBUG_STAMPS
// Diagnostic build only (libmi_instance_pipeline_dbg.so): copy out the per-tile stamps.
int32_t mip_debug_read_stamps(MipContext* ctx, unsigned long long* out, uint32_t n_tiles) {
  if (!ctx || !out) return MIP_ERR_INVALID_ARGUMENT;
  MIP_HIP(ctx, hipSetDevice(ctx->device));
  MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MIP_HIP(ctx, hipMemcpy(out, ctx->d_stamps, (size_t)n_tiles * 64, hipMemcpyDeviceToHost));
  return MIP_OK;
}
int32_t mip_debug_write_stamps(MipContext* ctx, const unsigned long long* in, uint32_t n_tiles) {
  if (!ctx || !in) return MIP_ERR_INVALID_ARGUMENT;
  MIP_HIP(ctx, hipSetDevice(ctx->device));
  MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MIP_HIP(ctx, hipMemcpy(ctx->d_stamps, in, (size_t)n_tiles * 64, hipMemcpyHostToDevice));
  return MIP_OK;
}
#endif

}  // extern "C"
