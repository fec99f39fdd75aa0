// mip_api.hip — C ABI of the instance pipeline (include/mi_instance_pipeline.h) over the
// gfx950 kernels in instance_pipeline_kernels.hpp. HIP runtime only: no torch types, no
// CPU fallback. Modelled on the reference's one FFI precedent, the vma crate
// (vma/src/lib.rs:31-64; status-code returns as in src/renderer/device/alloc.rs:192-226).
#include "../../include/mi_instance_pipeline.h"
#include "instance_pipeline_kernels.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is opened with dlopen, never linked

#include <dlfcn.h>
#include <drm/drm.h>  // DRM sync objects: what an exported Vulkan semaphore fd is on amdgpu (kernel uapi, no libdrm)
#include <fcntl.h>
#include <sys/ioctl.h>
#include <time.h>
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

static_assert(sizeof(MipDrawIndexedIndirectCommand) == 20, "VkDrawIndexedIndirectCommand is 20 bytes");
static_assert(sizeof(MipMesh) == 80, "MipMesh layout");
static_assert(sizeof(MipShardHeader) == 32, "MipShardHeader layout");
static_assert(sizeof(mip::MeshEntry) == 32, "MeshEntry layout");
static_assert(sizeof(mip::KernelArgs) <= 4096, "kernel argument block");

struct MipContext {
  int device = -1;
  uint32_t max_instances = 0, max_meshes = 0, cfg_flags = 0;
  uint32_t n = 0, m = 0;
  bool have_instances = false, have_meshes = false;
  bool ordered_tiles = false;  // MIP_CFG_ORDERED_TILES, or set by the first MIP_ERR_TIMEOUT
  int force_order = 0;         // tuning (MIP_TUNE_ORDER): 1 or 3, 0 = by instance count
  bool force_general = false;  // tuning/tests (MIP_TUNE_FORCE_GENERAL): always launch the kernel with the literal cold path
  // One slot per frame in flight: its own stream and its own cross-tile prefix state, so that
  // consecutive frames may overlap on the device (MipConfig.frames_in_flight).
  struct FrameSlot {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    unsigned long long* d_status = nullptr;  // level-0 granules, accumulators, group starts
    uint32_t* d_scalars = nullptr;           // [0] draw_count, [1] index_total (host-output runs), [2] pre-triangle count, [3] command ticket, [5] tile ticket (ordered tiles)
    uint32_t* d_tmp_cmds = nullptr;          // per-triangle stage: the instance kernel's list before re-compaction
    uint32_t* d_tmp_src = nullptr;           //                     and each command's source index offset
    uint32_t* d_tmp_blocks = nullptr;        //                     re-compaction of large frames: one word per 1024 commands
    unsigned long long* d_part_status = nullptr;  //                small frames: one granule per (command, part)
    uint32_t tri_epoch = 0;                  //                     tag of the last parts launch on this slot
    bool parts_dirty = false;                //                     a parts launch timed out: clear the granules before the next one
    float* d_skin_box = nullptr;             // skinned frames: per instance posed mesh-space box {min xyz, -, max xyz, -}
    uint2* d_tile_agg = nullptr;             // ordered tiles, large launches (three launches): per tile {count, sum};
    uint2* d_group_prefix = nullptr;         //                                                   per group of 64 tiles the exclusive prefix;
    uint32_t* d_vis_scratch = nullptr;       //                                                   the frame's visibility bitmap when the caller did not ask for it
    // recorded launches (mip_run_many): the frames of one replay, read by the kernels (KernelArgs.frame_ring),
    // refreshed before every replay from one of two pinned staging halves
    uint32_t* d_frame_ring = nullptr;
    uint32_t* h_frame_stage = nullptr;
    uint32_t frame_ring_frames = 0, stage_next = 0;
    hipEvent_t stage_free[2] = {nullptr, nullptr};
    uint32_t epoch = 0;         // highest tag handed out on this state
    uint32_t last_tag = 0;      // tag of the last launch (what the level-0 words hold now)
    uint32_t zero_buf = 2;      // which accumulator buffer is all-zero now: 0, 1, or 2 = both
    bool status_dirty = false;  // instance count changed: clear the prefix state before the next launch
    // the frame issued last on this slot, kept so that a frame whose launch ran into MIP_ERR_TIMEOUT can be issued again
    // in the mode the context has switched to (recover_from_timeout)
    struct Replay {
      uint32_t issued = 0;  // frames issued on this slot since the streams were last drained
      MipFrame frame;
      MipOutputs out;
      bool skinned = false;
      void* palette = nullptr;
    } replay;
  };
  // mip_run_many replays: per slot one linear hipGraph of `frames` launches with baked tags
  // base_epoch+1 .. base_epoch+frames (see run_many_graphed).
  struct FrameGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint32_t base_epoch = 0;
  };
  struct GraphSet {
    std::vector<MipOutputs> outs;
    uint32_t first_slot = 0, frames_per_slot = 0;
    uint64_t generation = 0;
    std::vector<FrameGraph> per_slot;
  };
  std::vector<GraphSet> graph_sets;   // small LRU, newest last
  uint64_t graph_generation = 1;      // bumped whenever something a graph bakes in changes
  uint32_t graph_round = 64;          // frames per replay round over all slots (MIP_TUNE_GRAPH_ROUND, 0 = off)
  std::vector<FrameSlot> slots;
  uint32_t next_slot = 0;
  std::vector<FrameSlot> view_states;  // mip_run_views: one prefix state per view, all on `stream`
  hipStream_t stream = nullptr;  // = slots[0].stream: uploads, merges, timing
  // resident inputs
  float* d_pos = nullptr;
  float4* d_rot = nullptr;
  float* d_scale = nullptr;
  uint32_t* d_mesh_id = nullptr;
  mip::MeshEntry* d_meshes = nullptr;
  mip::MeshDraw* d_mesh_draw = nullptr;
  unsigned long long* d_blas = nullptr;  // per-mesh BLAS device addresses (optional, row f-4)
  // consolidated geometry for the per-triangle stage (row f-1)
  float* d_vertices = nullptr;
  uint32_t* d_indices = nullptr;
  uint32_t n_vertices = 0, n_indices = 0;
  bool have_geometry = false;
  bool geometry_finite = false;  // every uploaded position is finite (lets the triangle stage skip exact no-ops)
  // host copies kept to check the mesh table against the geometry before the per-triangle stage gathers
  // vertices[vertex_offset + index] and indices[index_offset ..] unchecked on the device
  std::vector<MipMesh> h_meshes;
  std::vector<uint32_t> h_indices;
  int geometry_checked = 0;  // 0 = not yet, 1 = consistent, -1 = inconsistent (message in geometry_error)
  std::string geometry_error;
  // skinned extension (mip_set_skeleton / mip_set_poses / mip_run_skinned)
  mip::JointEntry* d_joints = nullptr;
  uint32_t n_joints = 0, max_joint_depth = 0;
  float joint_box_bound = INFINITY;  // 3 * max |joint_box| + 1, +inf while a joint box holds a non-finite value (SkinArgs.box_bound)
  uint8_t joint_level_start[mip::kMaxJoints + 2] = {0};
  uint32_t joint_level_inv[mip::kMaxJoints + 1] = {0};
  float* d_poses_owned = nullptr;
  const float* d_poses = nullptr;  // owned copy or a borrowed device pointer
  uint32_t poses_n = 0;
  int cu_count = 0;
  // layout of a slot's prefix state (words of 8 bytes)
  size_t status_bytes = 0;
  uint32_t acc1_offset_words = 0, start1_offset_words = 0, groups_cap = 0;
  uint32_t lds_pad = 0;  // tuning only (MIP_TUNE_LDS_PAD): dynamic LDS bytes that cap workgroups per CU
  uint32_t tri_block_threads = 0;     // tuning (MIP_TUNE_TRI_BLOCK_THREADS): 256 / 512 / 1024, 0 = by instance count
  uint32_t ordered_three_pass_min_tiles = 160;   // ordered tiles: launches of more tiles than this (40 960 instances) take wait-free launches
                                                  // instead of one ticket per tile (MIP_TUNE_THREE_PASS_MIN_TILES; measured crossover
                                                  // in profiles/r03_ordered_tiles_three_pass.txt: 32 k instances 6.3 against 6.5 us, 65 k 7.7 against 6.9)
  uint32_t emit_self_prefix_tiles = mip::kEmitSelfPrefixTiles;  //   ... two launches up to this many tiles, three above (MIP_TUNE_EMIT_SELF_PREFIX_TILES; 0: always three)
  uint32_t tri_block_max = 65536;  // instance counts up to this use the workgroup-per-command triangle kernel
  uint32_t tri_parts_max = 1024;   // instance counts up to this use the parts kernel (16 work items per command), 0 = off
                                   // measured (DamagedHelmet entry, frame time parts / workgroup-per-command): 30 instances 14 / 24 us,
                                   // 200: 16 / 25, 1000: 41 / 47, 2000: 67 / 64, 4000: 113 / 83
  uint32_t max_lod_tris = 0;       // largest triangle count of LOD 0 / LOD 1 over the mesh table
  // upload-time census of instances that fail the kernel's finite test (instance_kernel.hpp,
  // finite_magnitude): while it is zero, frames run the kernel without the literal cold path
  uint64_t nonfinite_instances = 0;
  float box_abs = 0.f;  // largest sum of |box coordinates| over the mesh table (the census' overflow bound)
  uint32_t* d_census = nullptr;
  uint32_t* h_error = nullptr;  // pinned, device-visible
  uint32_t* d_error = nullptr;  // device alias of h_error
  // staging for MIP_OUT_HOST
  float4* s_model = nullptr;
  uint32_t* s_bitmap = nullptr;
  uint32_t* s_cmds = nullptr;
  float* s_aabb = nullptr;
  // timing
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  MipTimings timings{};
  bool pending_async = false;
  // native sharded exchange (mip_comm_*, mip_run_sharded)
  ncclComm_t comm = nullptr;
  uint32_t comm_rank = 0, comm_world = 0;
  uint32_t shard_cap_max = 0;  // largest max_instances over the ranks: what every rank sizes its chunks by
  bool replay_blocked = false; // since the streams were last drained something was issued that cannot be issued again from a record:
                               // a recorded round of mip_run_many, a multi-view or sharded frame, an external semaphore operation
  bool recovering = false;
  uint32_t last_error_bits = 0;
  int shard_wire = 2;          // what mip_run_sharded exchanges: 2 = the packed wire form whenever the largest shard fits it (else 1),
                               // 1 = 8-byte wire records, 0 = 20-byte commands (MIP_TUNE_SHARD_WIRE: A/B and tests)
  int sharded_form = 0;        // the form of the frame in flight (what the send buffer holds; repair_sharded_overflow re-sends it)
  uint32_t* d_send = nullptr;  // this rank's chunk, sized for max_instances commands
  uint32_t* d_recv = nullptr;  // world chunks
  // the last sharded frame, kept so that a tightened chunk that overflowed can be re-gathered at full capacity
  void* sharded_out_cmds = nullptr;
  uint32_t* sharded_out_count = nullptr;
  uint32_t sharded_pending = 0;  // sharded frames enqueued since the last completed wait
  // imported external memory (mip_import_external_fd)
  struct External { hipExternalMemory_t mem; void* ptr; };
  std::vector<External> externals;
  // imported external semaphores (mip_import_external_semaphore_fd); the handle given out is the entry's address
  // Two implementations behind one handle: the HIP runtime's own (hipImportExternalSemaphore: waits and signals
  // execute on the device), or — when the runtime refuses the handle type, as ROCm 7.2 on Linux does — the kernel
  // object itself: the fd of an exported Vulkan semaphore is a DRM sync object on amdgpu, imported on a render node
  // and waited for / signalled by host functions enqueued on the frame's stream (hipLaunchHostFunc).
  struct ExternalSemaphore { hipExternalSemaphore_t sem; uint32_t kind; uint32_t drm_handle; };
  std::vector<ExternalSemaphore*> semaphores;
  int drm_fd = -1;  // render node, opened on first use
  uint32_t last_slot = 0;  // slot of the frame issued last (mip_signal_external goes behind it)
  char err[512] = {0};
#ifdef MIP_DEBUG_STAMPS
  unsigned long long* d_stamps = nullptr;
#endif
};

namespace {

int32_t fail(MipContext* ctx, int32_t code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
    va_end(ap);
  }
  return code;
}

#define MIP_HIP(ctx, call)                                                                    \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(ctx, e_ == hipErrorOutOfMemory ? MIP_ERR_OUT_OF_MEMORY : MIP_ERR_DEVICE,     \
                  "%s failed: %s", #call, hipGetErrorString(e_));                             \
  } while (0)

uint32_t tiles_for(uint32_t n) { return (n + mip::kTile - 1) / mip::kTile; }

int32_t bind_device(MipContext* ctx) {
  MIP_HIP(ctx, hipSetDevice(ctx->device));
  return MIP_OK;
}

int32_t sync_all(MipContext* ctx) {
  for (auto& sl : ctx->slots) MIP_HIP(ctx, hipStreamSynchronize(sl.stream));
  return MIP_OK;
}

int32_t ensure_staging(MipContext* ctx, const MipOutputs* out) {
  const size_t cap = ctx->max_instances ? ctx->max_instances : 1;
  if (out->model && !ctx->s_model) MIP_HIP(ctx, hipMalloc(&ctx->s_model, cap * 64));
  if (out->visible_bitmap && !ctx->s_bitmap) MIP_HIP(ctx, hipMalloc(&ctx->s_bitmap, ((cap + 31) / 32) * 4));
  if (out->draw_cmds && !ctx->s_cmds) MIP_HIP(ctx, hipMalloc(&ctx->s_cmds, cap * 20));
  if (out->world_aabb && !ctx->s_aabb) MIP_HIP(ctx, hipMalloc(&ctx->s_aabb, cap * 24));
  return MIP_OK;
}

int32_t repair_sharded_overflow(MipContext* ctx);

// Reads and clears the device-visible error words (one per kind, instance_kernel.hpp) after the streams have drained.
//
// A sharded frame may need a COLLECTIVE repair (the all-gather + merge repeated at full capacity when a tightened
// chunk overflowed). Whether it does is decided from the gathered headers alone — kErrChunkOverflow is raised by the
// merge kernel, which sees the same headers on every rank — and never from anything only this rank knows: a rank
// whose own shard kernel timed out (half-written chunk, garbage header, likely an "overflow" everywhere) still takes
// part in the repair its peers are entering, and reports its local error afterwards. Returning before the repair
// (round 2 did) left the other ranks blocked in ncclAllGather for good.
int32_t check_device_error(MipContext* ctx) {
  uint32_t e = 0;
  for (uint32_t k = 0; k < mip::kErrWords; ++k) {
    e |= ((volatile uint32_t*)ctx->h_error)[k];
    ((volatile uint32_t*)ctx->h_error)[k] = 0;
  }
  ctx->last_error_bits = e;
  if (!e) return MIP_OK;
  int32_t repair_rc = MIP_OK;
  if (e & mip::kErrChunkOverflow) repair_rc = repair_sharded_overflow(ctx);
  if (e & mip::kErrTimeout) {
    // the frame's prefix state is half-written: clear it before the next launch, and from now on
    // number the tiles from a counter, which cannot stall on the order workgroups start in
    for (auto& sl : ctx->slots) sl.status_dirty = true;
    for (auto& sl : ctx->view_states) sl.status_dirty = true;
    ctx->ordered_tiles = true;
    ctx->graph_generation++;
    return fail(ctx, MIP_ERR_TIMEOUT,
                "prefix wait expired (an earlier tile never published); outputs invalid; the context now uses ordered tiles");
  }
  if (e & mip::kErrPartsTimeout) {
    // mip_triangle_cull_parts_kernel assumes its whole grid is resident (a part waits for the earlier parts of its
    // command, which other workgroups own); another tenant on the GPU can break that. Ordered tiles do not help
    // this kernel: stop using it — the workgroup-per-command kernel has no cross-workgroup wait.
    ctx->tri_parts_max = 0;
    for (auto& sl : ctx->slots) sl.parts_dirty = true;
    return fail(ctx, MIP_ERR_TIMEOUT,
                "a part of a command waited in vain for an earlier part (the parts kernel was not resident as a whole); outputs "
                "invalid; the context now culls triangles with one workgroup per command");
  }
  if (e & mip::kErrIndexOverflow)
    return fail(ctx, MIP_ERR_CAPACITY, "culled_index_buffer too small for a command's index range; its triangles were dropped");
  if (e & mip::kErrSemaphore)
    return fail(ctx, MIP_ERR_TIMEOUT, "a wait for (or signal of) an external semaphore failed or expired after 10 s; the frame behind it ran anyway");
  if (e & mip::kErrWireRecord)
    return fail(ctx, MIP_ERR_DEVICE, "a wire record names a mesh outside this context's mesh table (corrupt chunk, or the ranks hold different tables)");
  return repair_rc;
}

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
  ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
  ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*get_error_string)(ncclResult_t) = nullptr;
};

// RCCL is an optional dependency: resolved on first use. In a process that already has a
// librccl.so.1 (torch ships one) dlopen returns that copy.
const RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // The collective library is a seam: anything that exports ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
    // ncclAllGather (and optionally ncclGetErrorString) will do. MIP_COMM_LIBRARY names it; the default is RCCL.
    // (tests/fake_ccl is a shared-memory double with which the native sharded frame runs with several ranks on one GPU.)
    void* h = nullptr;
    if (const char* env = std::getenv("MIP_COMM_LIBRARY")) {
      h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    } else {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    }
    if (h) {
      api.get_unique_id = (decltype(api.get_unique_id))dlsym(h, "ncclGetUniqueId");
      api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(h, "ncclCommInitRank");
      api.comm_destroy = (decltype(api.comm_destroy))dlsym(h, "ncclCommDestroy");
      api.all_gather = (decltype(api.all_gather))dlsym(h, "ncclAllGather");
      api.get_error_string = (decltype(api.get_error_string))dlsym(h, "ncclGetErrorString");
      if (api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_gather) api.handle = h;
    }
  }
  return api.handle ? &api : nullptr;
}

// Which instantiation of the frame kernel a launch uses: the one without the literal cold path
// whenever the census says every resident instance passes the finite test (and no per-instance
// box override, which may be non-finite, is in play).
using FrameKernel = mip::FrameKernelFn;
// the stores-first order (kOrder == 1) is instantiated here; the commands-first order (3) in stages_tu.hip, built with
// other flags — never both in one unit (stage_args.hpp)
template <bool kBox, bool kGeneral, int kWire>
FrameKernel pick_order(bool ticketed, int order) {
  if (order != 1) return mip::frame_kernel_commands_first(ticketed, kBox, kGeneral, kWire);
  return ticketed ? (FrameKernel)mip::mip_instance_pipeline_kernel<true, kBox, kGeneral, 1, kWire>
                  : (FrameKernel)mip::mip_instance_pipeline_kernel<false, kBox, kGeneral, 1, kWire>;
}
int wire_form(uint32_t out_flags) { return (out_flags & MIP_OUT_WIRE) ? ((out_flags & MIP_OUT_WIRE_PACKED) ? 2 : 1) : 0; }
FrameKernel select_frame_kernel(const MipContext* ctx, bool box_override, int wire /* 0 | 1 | 2 = packed */, uint32_t* grid) {
  const uint32_t n_tiles = tiles_for(ctx->n);
  *grid = n_tiles;
  // order (instance_kernel.hpp): commands-first while the launch is less than about two generations of
  // workgroups (8 per CU: every tile is ramp or tail), stores-first once there is a steady state.
  // Measured on MI355X (profiles/r02_order*.txt, r02_next_generation_prefetch_ab.txt, r02_lookup_probes.txt):
  // order 3 ahead below ~0.8 M instances (200 k: 7.0 vs 7.9 us; 700 k: 13.6 vs 14.3), order 1 ahead from ~1 M
  // (2 M: 34.1 vs 35.7; 4 M: 64.4 vs 68; 10 M: 181 vs 232 us; 1 M: equal within the run-to-run spread).
  int order = n_tiles <= (uint32_t)ctx->cu_count * 14u ? 3 : 1;
  if (ctx->force_order) order = ctx->force_order;
  if (box_override) return pick_order<true, true, 0>(ctx->ordered_tiles, order);  // (validate_run refuses MIP_OUT_WIRE for skinned frames)
  const bool general = ctx->nonfinite_instances != 0 || ctx->force_general;
  if (wire == 2) return general ? pick_order<false, true, 2>(ctx->ordered_tiles, order) : pick_order<false, false, 2>(ctx->ordered_tiles, order);
  if (wire == 1) return general ? pick_order<false, true, 1>(ctx->ordered_tiles, order) : pick_order<false, false, 1>(ctx->ordered_tiles, order);
  return general ? pick_order<false, true, 0>(ctx->ordered_tiles, order) : pick_order<false, false, 0>(ctx->ordered_tiles, order);
}

// Number of instances of [first, first + count) of the resident columns that fail the finite test, and (bad_ids != null)
// how many of them name a mesh outside a table of `m` entries.
// Synchronous (uploads are): one small kernel and an 8-byte read-back on the upload stream.
int32_t census(MipContext* ctx, uint32_t first, uint32_t count, uint32_t* out, uint32_t* bad_ids = nullptr, uint32_t m = 0) {
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
  a.tile_ticket = sl.d_scalars + 5;
  a.status0 = sl.d_status;
  a.acc1 = sl.d_status + ctx->acc1_offset_words;
  a.start1 = sl.d_status + ctx->start1_offset_words;
  a.groups_cap = ctx->groups_cap;
  a.error_flag = ctx->d_error;
  a.n = n;
  a.bitmap_words = (n + 31u) / 32u;
  a.n_meshes = ctx->m;
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
  if (const char* env = std::getenv("MIP_DEBUG_SKIP_PUBLISH_TILE")) a.debug_skip_publish_tile = (uint32_t)std::atoi(env) + 1u;
  if (const char* env = std::getenv("MIP_DEBUG_SKIP_PUBLISH_ONCE")) {  // fault injection for the transparent recovery: the first launches only
    static int left = std::getenv("MIP_DEBUG_SKIP_PUBLISH_LAUNCHES") ? std::atoi(std::getenv("MIP_DEBUG_SKIP_PUBLISH_LAUNCHES")) : 1;
    if (left > 0) {
      --left;
      a.debug_skip_publish_tile = (uint32_t)std::atoi(env) + 1u;
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

int32_t validate_run(MipContext* ctx, const MipFrame* frame, const MipOutputs* out) {
  if (!frame || !out) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "frame/out is NULL");
  if (!ctx->have_instances || !ctx->have_meshes) return fail(ctx, MIP_ERR_NOT_READY, "instances or mesh table not set");
  if ((out->draw_cmds == nullptr) != (out->draw_count == nullptr))
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "draw_cmds and draw_count go together");
  if (out->draw_index_total && !out->draw_cmds)
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "draw_index_total needs draw_cmds");
  const bool device_out = (out->flags & MIP_OUT_DEVICE) != 0;
  if ((out->flags & MIP_OUT_ASYNC) && !device_out)
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_ASYNC needs MIP_OUT_DEVICE");
  if (out->flags & MIP_OUT_WIRE) {
    if (!device_out || !out->draw_cmds) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE needs MIP_OUT_DEVICE and draw_cmds");
    if (out->culled_index_buffer) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE cannot carry the per-triangle stage's indexCount");
    if (ctx->m > 0x7fffffffu) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE needs mesh ids below 2^31");
    if ((out->flags & MIP_OUT_WIRE_PACKED) && (uint64_t)ctx->n > (1ull << mip_wire_index_bits(ctx->m)))
      return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE_PACKED: %u instances do not fit the %u index bits a table of %u meshes leaves",
                  ctx->n, mip_wire_index_bits(ctx->m), ctx->m);
  } else if (out->flags & MIP_OUT_WIRE_PACKED) {
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE_PACKED goes with MIP_OUT_WIRE");
  }
  if (out->culled_index_buffer) {
    if (!device_out) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "culled_index_buffer needs MIP_OUT_DEVICE");
    if (!ctx->have_geometry) return fail(ctx, MIP_ERR_NOT_READY, "culled_index_buffer needs mip_set_geometry");
    if (!out->model || !out->draw_cmds) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "culled_index_buffer needs model and draw_cmds");
    if (int32_t rc = check_geometry(ctx)) return rc;
  }
  return MIP_OK;
}

void drop_graphs(MipContext* ctx) {
  for (auto& gs : ctx->graph_sets)
    for (auto& fg : gs.per_slot) {
      if (fg.exec) (void)hipGraphExecDestroy(fg.exec);
      if (fg.graph) (void)hipGraphDestroy(fg.graph);
    }
  ctx->graph_sets.clear();
}

void free_all(MipContext* ctx) {
  if (!ctx) return;
  if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
  for (auto& sl : ctx->slots)
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
  drop_graphs(ctx);
  for (auto& e : ctx->externals) (void)hipDestroyExternalMemory(e.mem);  // unmaps the buffer as well
  ctx->externals.clear();
  for (auto* e : ctx->semaphores) {
    if (e->sem) (void)hipDestroyExternalSemaphore(e->sem);
    if (e->drm_handle && ctx->drm_fd >= 0) {
      drm_syncobj_destroy d{};
      d.handle = e->drm_handle;
      (void)ioctl(ctx->drm_fd, DRM_IOCTL_SYNCOBJ_DESTROY, &d);
    }
    delete e;
  }
  ctx->semaphores.clear();
  if (ctx->drm_fd >= 0) close(ctx->drm_fd);
  ctx->drm_fd = -1;
  (void)hipFree(ctx->d_pos);
  (void)hipFree(ctx->d_rot);
  (void)hipFree(ctx->d_scale);
  (void)hipFree(ctx->d_mesh_id);
  (void)hipFree(ctx->d_meshes);
  (void)hipFree(ctx->d_census);
  if (ctx->comm && rccl()) (void)rccl()->comm_destroy(ctx->comm);
  (void)hipFree(ctx->d_send);
  (void)hipFree(ctx->d_recv);
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
    (void)hipFree(sl.d_part_status);
    (void)hipFree(sl.d_skin_box);
    (void)hipFree(sl.d_tile_agg);
    (void)hipFree(sl.d_group_prefix);
    (void)hipFree(sl.d_vis_scratch);
    (void)hipFree(sl.d_frame_ring);
    if (sl.h_frame_stage) (void)hipHostFree(sl.h_frame_stage);
    for (auto& e : sl.stage_free)
      if (e) (void)hipEventDestroy(e);
  }
  (void)hipFree(ctx->s_model);
  (void)hipFree(ctx->s_bitmap);
  (void)hipFree(ctx->s_cmds);
  (void)hipFree(ctx->s_aabb);
  if (ctx->h_error) (void)hipHostFree(ctx->h_error);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  for (auto& sl : ctx->slots)
    if (sl.own_stream && sl.stream) (void)hipStreamDestroy(sl.stream);
  delete ctx;
}

}  // namespace

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
  ctx->ordered_tiles = (cfg->flags & MIP_CFG_ORDERED_TILES) != 0;

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
    ctx->status_bytes = ((size_t)ctx->start1_offset_words + (size_t)ctx->groups_cap * 2) * 8;
    for (auto& sl : ctx->slots) {
      MIP_HIP(ctx, hipMalloc(&sl.d_status, ctx->status_bytes));
      MIP_HIP(ctx, hipMemset(sl.d_status, 0, ctx->status_bytes));  // epoch 0 is never used
      MIP_HIP(ctx, hipMalloc(&sl.d_scalars, 64));
      MIP_HIP(ctx, hipMemset(sl.d_scalars, 0, 64));
    }
    MIP_HIP(ctx, hipHostMalloc(&ctx->h_error, 64, hipHostMallocMapped));
    std::memset(ctx->h_error, 0, 64);
    MIP_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->d_error, ctx->h_error, 0));
#ifdef MIP_DEBUG_STAMPS
    MIP_HIP(ctx, hipMalloc(&ctx->d_stamps, tiles_cap * 64));
    MIP_HIP(ctx, hipMemset(ctx->d_stamps, 0, tiles_cap * 64));
#endif
    if (const char* env = std::getenv("MIP_TUNE_LDS_PAD")) ctx->lds_pad = (uint32_t)std::atoi(env);
    if (const char* env = std::getenv("MIP_TUNE_TRI_BLOCK_THREADS")) {
      const uint32_t v = (uint32_t)std::atoi(env);
      if (v == 256u || v == 512u || v == 1024u) ctx->tri_block_threads = v;
    }
    if (const char* env = std::getenv("MIP_TUNE_TRI_BLOCK_MAX")) ctx->tri_block_max = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("MIP_TUNE_TRI_PARTS_MAX")) ctx->tri_parts_max = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("MIP_TUNE_ORDERED_TILES")) ctx->ordered_tiles = std::atoi(env) != 0;
    if (const char* env = std::getenv("MIP_TUNE_THREE_PASS_MIN_TILES")) ctx->ordered_three_pass_min_tiles = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("MIP_TUNE_EMIT_SELF_PREFIX_TILES")) ctx->emit_self_prefix_tiles = (uint32_t)std::strtoul(env, nullptr, 10);
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
  if (ctx && ctx->pending_async) ctx->replay_blocked = true;  // resident state changes: a frame in flight is never re-issued against the new state
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
  if (ctx && ctx->pending_async) ctx->replay_blocked = true;  // resident state changes: a frame in flight is never re-issued against the new state
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
  if (ctx && ctx->pending_async) ctx->replay_blocked = true;  // resident state changes: a frame in flight is never re-issued against the new state
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
  if (ctx && ctx->pending_async) ctx->replay_blocked = true;  // resident state changes: a frame in flight is never re-issued against the new state
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
  if (ctx && ctx->pending_async) ctx->replay_blocked = true;  // resident state changes: a frame in flight is never re-issued against the new state
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

static int32_t recover_from_timeout(MipContext* ctx, int32_t rc);

static int32_t run_frame(MipContext* ctx, const MipFrame* frame, const MipOutputs* out, bool skinned, void* palette) {
  if (int32_t rc = validate_run(ctx, frame, out)) return rc;
  if (skinned && (out->flags & MIP_OUT_WIRE)) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "MIP_OUT_WIRE is not available for skinned frames");
  const bool device_out = (out->flags & MIP_OUT_DEVICE) != 0;
  const bool async = device_out && (out->flags & MIP_OUT_ASYNC) != 0;
  const bool triangles = out->culled_index_buffer != nullptr;
  if (int32_t rc = bind_device(ctx)) return rc;

  // Frames rotate over the slots; a slot's stream orders a frame after the frame that last
  // used the same prefix state.
  MipContext::FrameSlot& sl = ctx->slots[ctx->next_slot];
  ctx->last_slot = ctx->next_slot;
  ctx->next_slot = (ctx->next_slot + 1) % (uint32_t)ctx->slots.size();
  hipStream_t stream = sl.stream;
  const bool alone = !ctx->pending_async;  // nothing else of this context is in flight: an error seen at the end of a synchronous frame is its own
  const uint32_t this_slot = ctx->last_slot;
  if (async) {  // on record until the streams are drained (a synchronous frame is repeated from its live arguments instead: below)
    sl.replay.issued += 1;
    sl.replay.frame = *frame;
    sl.replay.out = *out;
    sl.replay.skinned = skinned;
    sl.replay.palette = palette;
  }

  const uint32_t n = ctx->n;
  const uint32_t words = (n + 31u) / 32u;
  if (!device_out)
    if (int32_t rc = ensure_staging(ctx, out)) return rc;

  mip::KernelArgs a{};
  fill_kernel_args(ctx, sl, frame, out, device_out, a);
  if (out->tlas_instances && !device_out) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "tlas_instances needs MIP_OUT_DEVICE");
  if (triangles) {
    // the instance kernel emits into the slot's scratch list; the triangle stage rewrites
    // indexCount there and the final compaction lands in the caller's buffers
    const size_t cap = ctx->max_instances ? ctx->max_instances : 1;
    if (!sl.d_tmp_cmds) MIP_HIP(ctx, hipMalloc(&sl.d_tmp_cmds, cap * 20));
    if (!sl.d_tmp_src) MIP_HIP(ctx, hipMalloc(&sl.d_tmp_src, cap * 4));
    if (!sl.d_tmp_blocks) MIP_HIP(ctx, hipMalloc(&sl.d_tmp_blocks, (cap / 1024 + 1) * 4));
    a.cmds = sl.d_tmp_cmds;
    a.draw_count = sl.d_scalars + 2;
    a.src_index_offset = sl.d_tmp_src;
  }
  if (skinned) {
    // the posed mesh-space box replaces the mesh table's; computed first, on the same stream
    if (!sl.d_skin_box) MIP_HIP(ctx, hipMalloc(&sl.d_skin_box, (size_t)(ctx->max_instances ? ctx->max_instances : 1) * 32));
    a.box_override = sl.d_skin_box;  // per frame slot: frames in flight may carry different poses
  }

  // Cross-tile prefix state (see instance_kernel.hpp): a fresh tag per launch marks
  // the level-0 words; the level-1 accumulators alternate between two buffers by tag parity,
  // the kernel zeroing the other one. Launches without draw commands do not touch the state.
  if (a.cmds && n) {
    if (int32_t rc = reset_prefix_state_if_needed(ctx, sl, 2)) return rc;
    uint32_t e = sl.epoch + 1;
    if (sl.zero_buf != 2 && (e & 1u) != sl.zero_buf) ++e;  // must accumulate in the zeroed buffer
    a.epoch = sl.epoch = sl.last_tag = e;
    sl.zero_buf = (e & 1u) ^ 1u;
  }

  const bool timing = (ctx->cfg_flags & MIP_CFG_TIMING) != 0;
  if (n == 0) {
    if (triangles) MIP_HIP(ctx, hipMemsetAsync(out->draw_count, 0, 4, stream));
    if (a.draw_count) MIP_HIP(ctx, hipMemsetAsync(a.draw_count, 0, 4, stream));
    if (a.index_total) MIP_HIP(ctx, hipMemsetAsync(a.index_total, 0, 4, stream));
  } else {
    if (timing) MIP_HIP(ctx, hipEventRecord(ctx->ev0, stream));
    if (skinned) {
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
      const uint32_t per_block = 4u * (64u / ctx->n_joints);
      mip::launch_skinned_bounds((n + per_block - 1) / per_block, stream, k);
      MIP_HIP(ctx, hipGetLastError());
    }
    bool frame_launched = false;
    if (a.cmds && ctx->ordered_tiles && a.n_tiles > ctx->ordered_three_pass_min_tiles) {
      // ordered tiles, large launch: three launches none of which waits for another workgroup
      // (instance_kernel.hpp, "the prefix without any wait"; emit_kernel.hpp)
      const size_t tiles_cap = tiles_for(ctx->max_instances ? ctx->max_instances : 1);
      const size_t groups_cap = (tiles_cap + mip::kTileGroup - 1) / mip::kTileGroup;
      if (!sl.d_tile_agg) MIP_HIP(ctx, hipMalloc(&sl.d_tile_agg, tiles_cap * sizeof(uint2)));
      if (!sl.d_group_prefix) MIP_HIP(ctx, hipMalloc(&sl.d_group_prefix, groups_cap * sizeof(uint2)));
      // 1: the frame kernel without commands — matrices, boxes, TLAS rows, the visibility bitmap — leaves one pair per tile
      if (!a.bitmap) {  // the caller did not ask for the bitmap: launch 3 still needs it
        if (!sl.d_vis_scratch) MIP_HIP(ctx, hipMalloc(&sl.d_vis_scratch, ((size_t)(ctx->max_instances ? ctx->max_instances : 1) + 31) / 32 * 4));
        a.bitmap = sl.d_vis_scratch;
      }
      mip::KernelArgs a1 = a;
      a1.cmds = nullptr; a1.draw_count = nullptr; a1.index_total = nullptr; a1.src_index_offset = nullptr;
      a1.tile_agg_out = sl.d_tile_agg;
      void* params[1] = {&a1};
      if (skinned || ctx->nonfinite_instances != 0 || ctx->force_general) ctx->timings.general_launches += 1;
      uint32_t grid = 0;
      const FrameKernel kernel = select_frame_kernel(ctx, skinned, 0, &grid);  // ordered tiles: the ticketed instantiation
      MIP_HIP(ctx, hipLaunchKernel((const void*)kernel, dim3(grid), dim3(mip::kTile), params, ctx->lds_pad, stream));
      frame_launched = true;
      // 2: exclusive prefixes of the group sums + the totals — only for launches too large for launch 3 to sum the pairs itself
      const bool scan_launch = a.n_tiles > ctx->emit_self_prefix_tiles;
      if (scan_launch) {
        mip::TileScanArgs ts{};
        ts.tile_agg = sl.d_tile_agg;
        ts.group_prefix = sl.d_group_prefix;
        ts.n_tiles = a.n_tiles;
        ts.draw_count = a.draw_count;
        ts.index_total = a.index_total;
        hipLaunchKernelGGL(mip::mip_tile_scan_kernel, dim3(1), dim3(1024), 0, stream, ts);
        MIP_HIP(ctx, hipGetLastError());
      }
      // 3: the commands, from the bitmap and the prefixes
      mip::EmitArgs e{};
      e.pos = a.pos; e.mesh_id = a.mesh_id; e.meshes = a.meshes; e.mesh_draw = a.mesh_draw;
      e.bitmap = a.bitmap;
      e.tile_agg = sl.d_tile_agg;
      e.group_prefix = scan_launch ? sl.d_group_prefix : nullptr;
      e.draw_count = a.draw_count;
      e.index_total = a.index_total;
      e.n_tiles = a.n_tiles;
      e.cmds = a.cmds;
      e.src_index_offset = a.src_index_offset;
      e.n = n;
      e.first_instance_base = a.first_instance_base;
      e.first_index_base = a.first_index_base;
      e.wire_index_bits = a.wire_index_bits;
      std::memcpy(e.cam, a.cam, sizeof e.cam);
      mip::launch_emit_commands(device_out ? wire_form(out->flags) : 0, a.n_tiles, stream, e);
      MIP_HIP(ctx, hipGetLastError());
      ctx->timings.three_pass_frames += 1;
    }
    if (!frame_launched) {
      void* params[1] = {&a};
      if (skinned || ctx->nonfinite_instances != 0 || ctx->force_general) ctx->timings.general_launches += 1;
      uint32_t grid = 0;
      const FrameKernel kernel = select_frame_kernel(ctx, skinned, device_out ? wire_form(out->flags) : 0, &grid);
      MIP_HIP(ctx, hipLaunchKernel((const void*)kernel, dim3(grid), dim3(mip::kTile), params, ctx->lds_pad, stream));
    }
    if (triangles) {
      mip::TriangleArgs t{};
      t.cmds = sl.d_tmp_cmds;
      t.count = sl.d_scalars + 2;
      t.src_index_offset = sl.d_tmp_src;
      t.model = (const float4*)out->model;
      t.vertices = ctx->d_vertices;
      t.indices = ctx->d_indices;
      t.out_indices = (uint32_t*)out->culled_index_buffer;
      t.capacity = out->culled_index_capacity;
      t.first_instance_base = frame->first_instance_base;
      t.error_flag = ctx->d_error;
      t.ticket = sl.d_scalars + 3;
      t.geometry_finite = ctx->geometry_finite ? 1u : 0u;
      std::memcpy(t.pv, frame->pv, sizeof t.pv);
      // The command count lives on the device; the instance count bounds it. Small frames: one
      // 1024-thread workgroup per command; large frames: one wave per command (no barriers).
      // (the parts kernel waits across workgroups and needs its whole grid resident: only while this context runs
      //  one frame at a time — two slots' launches could each be half resident; see kErrPartsTimeout for other tenants)
      const bool parts = ctx->tri_parts_max && n <= ctx->tri_parts_max && !ctx->tri_block_threads && ctx->slots.size() == 1 &&
                         ctx->max_lod_tris <= mip::kTriParts * 256u * mip::kTriPartMaxT;
      if (parts) {
        // small frames: 16 parts per command, handed out in order by a ticket counter (triangle_kernels.hpp)
        const size_t cap_cmds = ctx->max_instances < ctx->tri_parts_max ? (ctx->max_instances ? ctx->max_instances : 1) : ctx->tri_parts_max;
        if (!sl.d_part_status) {
          MIP_HIP(ctx, hipMalloc(&sl.d_part_status, cap_cmds * mip::kTriParts * 8));
          MIP_HIP(ctx, hipMemsetAsync(sl.d_part_status, 0, cap_cmds * mip::kTriParts * 8, stream));
          sl.tri_epoch = 0;
        }
        if (sl.tri_epoch == 0xffffffffu || sl.parts_dirty) {  // tag wrap (or a timed-out launch): start over on a cleared array
          sl.parts_dirty = false;
          MIP_HIP(ctx, hipMemsetAsync(sl.d_part_status, 0, cap_cmds * mip::kTriParts * 8, stream));
          sl.tri_epoch = 0;
        }
        mip::TrianglePartsArgs pa{};
        pa.t = t;
        pa.part_status = sl.d_part_status;
        pa.epoch = ++sl.tri_epoch;
        uint32_t blocks = n * mip::kTriParts;
        const uint32_t max_blocks = (uint32_t)ctx->cu_count * 4u;  // resident as a whole at this kernel's 121 VGPRs (4 waves per SIMD)
        if (blocks > max_blocks) blocks = max_blocks;
        mip::launch_triangle_cull_parts(blocks, stream, pa);
      } else if (n <= ctx->tri_block_max) {
        // workgroup size: the register budget allows 16 waves per CU, so 1024 / 512 / 256 threads = 1 / 2 / 4
        // workgroups per CU; smaller workgroups wait less at the per-step barrier, larger ones finish a
        // lone command sooner
        // measured (DamagedHelmet table entry, frame time in us at 256 / 512 / 1024 threads): 200 instances
        // 35 / 28 / 27, 1000: 60 / 49 / 56, 2000: 72 / 66 / 73, 4000: 85 / 88 / 108, 20 k: 266 / 324 / 347;
        // from ~65 k instances wave-per-command wins (100 k: 1.15 vs 1.21 ms)
        const uint32_t tb = ctx->tri_block_threads ? ctx->tri_block_threads : (n <= 768u ? 1024u : (n <= 3072u ? 512u : 256u));
        const uint32_t per_cu = 2u * (1024u / tb);
        uint32_t blocks = n < (uint32_t)ctx->cu_count * per_cu ? n : (uint32_t)ctx->cu_count * per_cu;
        if (!blocks) blocks = 1u;
        mip::launch_triangle_cull_block(tb, blocks, stream, t);
      } else {
        MIP_HIP(ctx, hipMemsetAsync(t.ticket, 0, 4, stream));  // only the wave-per-command kernel hands out tickets
        uint32_t blocks = (n + 3u) / 4u;
        const uint32_t max_blocks = (uint32_t)ctx->cu_count * 8u;
        if (blocks > max_blocks) blocks = max_blocks;
        mip::launch_triangle_cull_waves(blocks, stream, t);
      }
      MIP_HIP(ctx, hipGetLastError());
      // (re-compacting inside the workgroup kernels, by the last workgroup to finish, was measured: the
      // agent-scope fences it needs cost more than the launch they save — 1 k instances 65 vs 49 us)
      if (n <= ctx->tri_block_max) {
        mip::RecompactArgs r{};
        r.in_cmds = sl.d_tmp_cmds;
        r.in_count = sl.d_scalars + 2;
        r.out_cmds = (uint32_t*)out->draw_cmds;
        r.out_count = out->draw_count;
        mip::launch_recompact(stream, r);
      } else {  // many commands: counts per 1024, one block scans them, scatter
        mip::RecompactWideArgs r{};
        r.in_cmds = sl.d_tmp_cmds;
        r.in_count = sl.d_scalars + 2;
        r.out_cmds = (uint32_t*)out->draw_cmds;
        r.out_count = out->draw_count;
        r.block_base = sl.d_tmp_blocks;
        r.n_blocks = (n + 1023u) / 1024u;
        mip::launch_recompact_wide(stream, r);
      }
      MIP_HIP(ctx, hipGetLastError());
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
  int32_t rc = check_device_error(ctx);
  if (alone && !ctx->recovering) {
    // a synchronous frame with nothing else in flight: a stall was its own, and nobody can have consumed its outputs yet — the
    // context has switched modes in check_device_error, the frame is issued once more from the caller's (still live) arguments
    constexpr uint32_t kRecoverable = mip::kErrTimeout | mip::kErrPartsTimeout;
    if (rc == MIP_ERR_TIMEOUT && (ctx->last_error_bits & kRecoverable) && !(ctx->last_error_bits & ~kRecoverable)) {
      ctx->recovering = true;
      ctx->next_slot = this_slot;
      rc = run_frame(ctx, frame, out, skinned, palette);
      ctx->recovering = false;
      if (rc == MIP_OK) ctx->timings.timeout_recoveries += 1;
    }
    for (auto& s2 : ctx->slots) s2.replay.issued = 0;
    ctx->replay_blocked = false;
  }
  return rc;
}

// A frame kernel's bounded wait has expired (kErrTimeout: the context has just switched itself to ordered tiles; kErrPartsTimeout: the
// parts kernel of the per-triangle stage is off from now on), seen by mip_wait. The ASYNCHRONOUS frames that were in flight are on
// record, one per frame slot (their outputs are device memory the caller keeps alive until mip_wait): they are issued again — in the mode that cannot stall — and the caller gets their results instead of MIP_ERR_TIMEOUT (counted in
// MipTimings.timeout_recoveries; the 0.5 s of the expired wait are the price). Not possible, and the error is reported as before,
// when a slot carried more than one frame since the streams were last drained, or when something was in flight that is not on
// record or has side effects elsewhere: a recorded round of mip_run_many, a multi-view or sharded frame (collective), a merge (it
// has consumed a frame's list), an external semaphore operation (its consumer may already have been released), an asynchronous
// frame on a caller-owned stream (the caller may have queued consumers behind it).
static int32_t recover_from_timeout(MipContext* ctx, int32_t rc) {
  constexpr uint32_t kRecoverable = mip::kErrTimeout | mip::kErrPartsTimeout;
  const uint32_t bits = ctx->last_error_bits;
  if (rc != MIP_ERR_TIMEOUT || ctx->recovering || ctx->replay_blocked || !(bits & kRecoverable) || (bits & ~kRecoverable)) return rc;
  for (auto& sl : ctx->slots) {
    if (sl.replay.issued > 1) return rc;
    // an asynchronous frame on a stream the CALLER owns (MipConfig.stream) may have consumers queued behind it that the library
    // cannot see (renderer_amd/sharded.py: the all-gather of the frame's list): they have run on the invalid result
    if (sl.replay.issued == 1 && (sl.replay.out.flags & MIP_OUT_ASYNC) && !sl.own_stream) return rc;
  }
  ctx->recovering = true;
  const uint32_t keep_next = ctx->next_slot, keep_last = ctx->last_slot;
  int32_t again = MIP_OK;
  for (size_t k = 0; k < ctx->slots.size() && again == MIP_OK; ++k) {
    MipContext::FrameSlot::Replay r = ctx->slots[k].replay;  // a copy: run_frame overwrites the record
    if (r.issued != 1) continue;
    ctx->next_slot = (uint32_t)k;
    // (records are asynchronous device-output frames: all slots first, one drain below)
    again = run_frame(ctx, &r.frame, &r.out, r.skinned, r.palette);
  }
  ctx->next_slot = keep_next;
  ctx->last_slot = keep_last;
  if (again == MIP_OK) again = sync_all(ctx);
  if (again == MIP_OK) again = check_device_error(ctx);
  ctx->pending_async = false;
  ctx->recovering = false;
  if (again == MIP_OK) ctx->timings.timeout_recoveries += 1;
  return again;
}

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
          void* params[1] = {&a};
          hipKernelNodeParams kp{};
          uint32_t grid = 0;
          kp.func = (void*)select_frame_kernel(ctx, false, wire_form(out->flags), &grid);
          kp.gridDim = dim3(grid);
          kp.blockDim = dim3(mip::kTile);
          kp.sharedMemBytes = ctx->lds_pad;
          kp.kernelParams = params;
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
      ctx->replay_blocked = true;  // a recorded round is not on the per-slot record: no transparent recovery from a timeout
      MIP_HIP(ctx, hipGraphLaunch(fg.exec, sl.stream));
      sl.last_tag = fg.base_epoch + G;
      if (sl.epoch < sl.last_tag) sl.epoch = sl.last_tag;
      sl.zero_buf = first_buf;  // G is even: the last launch zeroed the buffer the first one uses
      ctx->timings.graph_frames += G;
    }
  ctx->pending_async = true;
  return MIP_OK;
}

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
  // (ordered tiles on a large launch is three launches per frame, set up by run_frame: not recorded)
  const bool three_pass = ctx->ordered_tiles && tiles_for(ctx->n) > ctx->ordered_three_pass_min_tiles;
  if (plain && !three_pass && ctx->graph_round && ctx->n && !(ctx->cfg_flags & MIP_CFG_TIMING)) {
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
          if (int32_t rc = validate_run(ctx, &frames[0], &outputs[k])) return rc;
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
  ctx->replay_blocked = true;  // a multi-view launch is not on the per-slot record
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
  if (ctx->ordered_tiles) {
    // The multi-view kernel numbers its tiles by blockIdx.x. A context in ordered-tiles mode (MIP_CFG_ORDERED_TILES,
    // or after a MIP_ERR_TIMEOUT) must not depend on dispatch order: a view IS a frame, so run one ticketed frame
    // per view on the first stream — same bytes, n_views launches.
    for (uint32_t v = 0; v < n_views; ++v) {
      const MipOutputs& o = outs[v];
      if (!(o.flags & MIP_OUT_DEVICE) || !o.draw_cmds || !o.draw_count || o.model || o.world_aabb || o.tlas_instances || o.culled_index_buffer)
        return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "view %u: mip_run_views takes MIP_OUT_DEVICE outputs with visible_bitmap, draw_cmds, draw_count, draw_index_total only", v);
      MipOutputs one = o;
      one.flags = MIP_OUT_DEVICE | MIP_OUT_ASYNC;
      ctx->next_slot = 0;
      if (int32_t rc = run_frame(ctx, &frames[v], &one, false, nullptr)) return rc;
    }
    ctx->next_slot = 0;
    if (async) return MIP_OK;
    return mip_wait(ctx);
  }
  // four views per launch (one wave of a workgroup finishes one view); more views are more launches on the same stream
  for (uint32_t first = 0; first < n_views; first += mip::kMaxViews) {
    const uint32_t k = n_views - first < mip::kMaxViews ? n_views - first : mip::kMaxViews;
    const bool last = first + k == n_views;
    if (int32_t rc = run_views_chunk(ctx, frames + first, outs + first, k, async || !last)) return rc;
  }
  return MIP_OK;
}

int32_t mip_set_skeleton(MipContext* ctx, const int32_t* parent, const float* inverse_bind, const float* joint_box,
                         uint32_t n_joints) {
  if (ctx && ctx->pending_async) ctx->replay_blocked = true;  // resident state changes: a frame in flight is never re-issued against the new state
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
  if (ctx && ctx->pending_async) ctx->replay_blocked = true;  // resident state changes: a frame in flight is never re-issued against the new state
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
  ctx->pending_async = false;
  int32_t rc = check_device_error(ctx);
  ctx->sharded_pending = 0;
  if (rc == MIP_ERR_TIMEOUT) rc = recover_from_timeout(ctx, rc);
  for (auto& sl : ctx->slots) sl.replay.issued = 0;
  ctx->replay_blocked = false;
  return rc;
}

static int32_t enqueue_merge(MipContext* ctx, const void* chunks, uint32_t n_chunks, uint64_t chunk_stride_bytes,
                             uint32_t chunk_capacity, void* out_cmds, uint32_t* out_count) {
  ctx->replay_blocked = true;  // a merge consumes frame outputs: a frame behind it is never silently issued again
  mip::MergeArgs a{};
  a.chunks = (const unsigned char*)chunks;
  a.stride = chunk_stride_bytes;
  a.n_chunks = n_chunks;
  const uint64_t fits = (chunk_stride_bytes - sizeof(MipShardHeader)) / 20u;
  a.capacity = (chunk_capacity && chunk_capacity < fits) ? chunk_capacity : (uint32_t)(fits > 0xffffffffull ? 0xffffffffull : fits);
  a.out_cmds = (uint32_t*)out_cmds;
  a.out_count = out_count;
  a.error_flag = ctx->d_error;
  // Sized for the payload the chunks can hold: every thread moves ~8 words.
  const uint64_t max_words = (uint64_t)a.capacity * 5u * n_chunks;
  uint32_t blocks = (uint32_t)((max_words + 256 * 8 - 1) / (256 * 8));
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(mip::mip_merge_draw_lists_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
  MIP_HIP(ctx, hipGetLastError());
  return MIP_OK;
}

int32_t mip_merge_draw_lists(MipContext* ctx, const void* chunks, uint32_t n_chunks, uint64_t chunk_stride_bytes,
                             uint32_t chunk_capacity, void* out_cmds, uint32_t* out_count, int32_t async) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!chunks || !out_cmds || !out_count) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (n_chunks == 0 || n_chunks > mip::kMaxMergeChunks)
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "n_chunks %u outside 1..%u", n_chunks, mip::kMaxMergeChunks);
  if (chunk_stride_bytes < sizeof(MipShardHeader) || (chunk_stride_bytes & 3u))
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "bad chunk stride");
  if ((uint64_t)chunk_capacity * 20u + sizeof(MipShardHeader) > chunk_stride_bytes)
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "chunk_capacity %u does not fit a stride of %llu bytes", chunk_capacity,
                (unsigned long long)chunk_stride_bytes);
  if (int32_t rc = bind_device(ctx)) return rc;
  const bool timing = (ctx->cfg_flags & MIP_CFG_TIMING) != 0 && !async;
  if (timing) MIP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  if (int32_t rc = enqueue_merge(ctx, chunks, n_chunks, chunk_stride_bytes, chunk_capacity, out_cmds, out_count)) return rc;
  if (timing) MIP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  if (async) {
    ctx->pending_async = true;
    return MIP_OK;
  }
  MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (timing) {
    float ms = 0.f;
    MIP_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->timings.merges += 1;
    ctx->timings.last_merge_ms = ms;
    ctx->timings.total_merge_ms += ms;
  }
  return check_device_error(ctx);
}

static uint64_t wire_body_bytes(uint64_t capacity, bool packed) {
  return packed ? MIP_WIRE_PACKED_BODY_BYTES(capacity) : MIP_WIRE_BODY_BYTES(capacity);
}
static uint64_t wire_stride_bytes(uint64_t capacity, bool packed) {
  return (sizeof(MipShardHeader) + wire_body_bytes(capacity, packed) + 255) / 256 * 256;
}

static int32_t enqueue_merge_wire(MipContext* ctx, const void* chunks, uint32_t n_chunks, uint64_t chunk_stride_bytes,
                                  uint32_t chunk_capacity, void* out_cmds, uint32_t* out_count, bool packed) {
  static_assert(MIP_WIRE_BLOCK_COMMANDS == mip::kWireBlockCmds && MIP_WIRE_BLOCK_BYTES == mip::kWireBlockWords * 4u &&
                MIP_WIRE_BLOCK_HEADER_BYTES == mip::kWireBlockHeaderWords * 4u && MIP_WIRE_PACKED_BLOCK_BYTES == mip::kWirePackedBlockWords * 4u,
                "wire layout: header and kernels agree");
  ctx->replay_blocked = true;  // (as enqueue_merge)
  mip::MergeWireArgs a{};
  a.chunks = (const unsigned char*)chunks;
  a.stride = chunk_stride_bytes;
  a.n_chunks = n_chunks;
  a.capacity = chunk_capacity;
  a.out_cmds = (uint32_t*)out_cmds;
  a.out_count = out_count;
  a.error_flag = ctx->d_error;
  a.meshes = ctx->d_meshes;
  a.mesh_draw = ctx->d_mesh_draw;
  a.n_meshes = ctx->m;
  // one workgroup expands one block of 256 records at a time; sized for the blocks the chunks can hold
  const uint64_t max_blocks = ((uint64_t)chunk_capacity + mip::kWireBlockCmds - 1) / mip::kWireBlockCmds * n_chunks;
  const uint32_t grid_cap = std::getenv("MIP_TUNE_MERGE_GRID") ? (uint32_t)std::atoi(std::getenv("MIP_TUNE_MERGE_GRID")) : 256u * 8u;
  uint32_t blocks = max_blocks > grid_cap ? grid_cap : (uint32_t)max_blocks;
  if (blocks < 1) blocks = 1;
  if (packed) hipLaunchKernelGGL(mip::mip_merge_wire_lists_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream, a);
  else hipLaunchKernelGGL(mip::mip_merge_wire_lists_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, a);
  MIP_HIP(ctx, hipGetLastError());
  return MIP_OK;
}

static int32_t merge_wire_lists(MipContext* ctx, const void* chunks, uint32_t n_chunks, uint64_t chunk_stride_bytes,
                                uint32_t chunk_capacity, void* out_cmds, uint32_t* out_count, int32_t async, bool packed) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!chunks || !out_cmds || !out_count) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (!ctx->have_meshes) return fail(ctx, MIP_ERR_NOT_READY, "wire records are expanded against the mesh table: set it first");
  if (n_chunks == 0 || n_chunks > mip::kMaxMergeChunks)
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "n_chunks %u outside 1..%u", n_chunks, mip::kMaxMergeChunks);
  if (chunk_stride_bytes < sizeof(MipShardHeader)) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "bad chunk stride");
  if (chunk_capacity == 0) {  // what the stride holds, in whole blocks
    const uint64_t fits = (chunk_stride_bytes - sizeof(MipShardHeader)) / (packed ? MIP_WIRE_PACKED_BLOCK_BYTES : MIP_WIRE_BLOCK_BYTES) * MIP_WIRE_BLOCK_COMMANDS;
    chunk_capacity = fits > 0x3fffffffull ? 0x3fffffffu : (uint32_t)fits;
  }
  if ((chunk_stride_bytes & 15u) || sizeof(MipShardHeader) + wire_body_bytes(chunk_capacity, packed) > chunk_stride_bytes)
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "a wire chunk for %u commands does not fit a stride of %llu bytes (or the stride is not 16-byte aligned)",
                chunk_capacity, (unsigned long long)chunk_stride_bytes);
  if (int32_t rc = bind_device(ctx)) return rc;
  const bool timing = (ctx->cfg_flags & MIP_CFG_TIMING) != 0 && !async;
  if (timing) MIP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  if (int32_t rc = enqueue_merge_wire(ctx, chunks, n_chunks, chunk_stride_bytes, chunk_capacity, out_cmds, out_count, packed)) return rc;
  if (timing) MIP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  if (async) {
    ctx->pending_async = true;
    return MIP_OK;
  }
  MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (timing) {
    float ms = 0.f;
    MIP_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->timings.merges += 1;
    ctx->timings.last_merge_ms = ms;
    ctx->timings.total_merge_ms += ms;
  }
  return check_device_error(ctx);
}

int32_t mip_merge_wire_lists(MipContext* ctx, const void* chunks, uint32_t n_chunks, uint64_t chunk_stride_bytes,
                             uint32_t chunk_capacity, void* out_cmds, uint32_t* out_count, int32_t async) {
  return merge_wire_lists(ctx, chunks, n_chunks, chunk_stride_bytes, chunk_capacity, out_cmds, out_count, async, false);
}

int32_t mip_merge_wire_lists_packed(MipContext* ctx, const void* chunks, uint32_t n_chunks, uint64_t chunk_stride_bytes,
                                    uint32_t chunk_capacity, void* out_cmds, uint32_t* out_count, int32_t async) {
  return merge_wire_lists(ctx, chunks, n_chunks, chunk_stride_bytes, chunk_capacity, out_cmds, out_count, async, true);
}

uint32_t mip_wire_index_bits(uint32_t n_meshes) {
  uint32_t mesh_bits = 0;
  while (mesh_bits < 31u && (1ull << mesh_bits) < n_meshes) ++mesh_bits;  // ceil(log2(n_meshes)); 0 for one mesh
  return 31u - mesh_bits;
}

int32_t mip_comm_unique_id(uint8_t out_id[MIP_COMM_ID_BYTES]) {
  if (!out_id) return MIP_ERR_INVALID_ARGUMENT;
  const RcclApi* r = rccl();
  if (!r) return MIP_ERR_DEVICE;
  static_assert(sizeof(ncclUniqueId) == MIP_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  if (r->get_unique_id(&id) != ncclSuccess) return MIP_ERR_DEVICE;
  std::memcpy(out_id, &id, sizeof id);
  return MIP_OK;
}

int32_t mip_comm_init(MipContext* ctx, const uint8_t id[MIP_COMM_ID_BYTES], uint32_t rank, uint32_t world) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!id || world == 0 || rank >= world || world > mip::kMaxMergeChunks)
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "bad communicator arguments (rank %u of %u)", rank, world);
  if (ctx->slots.size() != 1) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "the sharded exchange needs frames_in_flight = 1");
  if (ctx->comm) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "communicator already initialised");
  const RcclApi* r = rccl();
  if (!r) {
    const char* why = dlerror();  // one call: dlerror() clears the message it returns
    return fail(ctx, MIP_ERR_DEVICE, "the collective library (MIP_COMM_LIBRARY or librccl.so.1) could not be loaded: %s", why ? why : "symbols missing");
  }
  if (int32_t rc = bind_device(ctx)) return rc;
  ncclUniqueId nid;
  std::memcpy(&nid, id, sizeof nid);
  const ncclResult_t res = r->comm_init_rank(&ctx->comm, (int)world, nid, (int)rank);
  if (res != ncclSuccess) {
    ctx->comm = nullptr;
    return fail(ctx, MIP_ERR_DEVICE, "ncclCommInitRank failed: %s", r->get_error_string ? r->get_error_string(res) : "?");
  }
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  if (const char* env = std::getenv("MIP_TUNE_SHARD_WIRE")) {  // A/B and tests: 0 = 20-byte commands, 1 = 8-byte records, 2 = packed when possible
    const int v = std::atoi(env);
    ctx->shard_wire = v < 0 ? 0 : (v > 2 ? 2 : v);
  }
  const int32_t rc = [&]() -> int32_t {
    // Chunks must have the same size on every rank, but ranks may have been created for different capacities (the
    // last of ceil(N/R)-sized shards is shorter): one 4-byte all-gather settles on the largest max_instances.
    uint32_t* d_caps = nullptr;
    MIP_HIP(ctx, hipMalloc(&d_caps, (size_t)(world + 1) * 4));
    const uint32_t mine = ctx->max_instances ? ctx->max_instances : 1u;
    std::vector<uint32_t> caps(world, 0u);
    const int32_t rc2 = [&]() -> int32_t {
      MIP_HIP(ctx, hipMemcpyAsync(d_caps + world, &mine, 4, hipMemcpyHostToDevice, ctx->stream));
      const ncclResult_t r2 = r->all_gather(d_caps + world, d_caps, 1, ncclUint32, ctx->comm, ctx->stream);
      if (r2 != ncclSuccess) return fail(ctx, MIP_ERR_DEVICE, "ncclAllGather failed: %s", r->get_error_string ? r->get_error_string(r2) : "?");
      MIP_HIP(ctx, hipMemcpyAsync(caps.data(), d_caps, (size_t)world * 4, hipMemcpyDeviceToHost, ctx->stream));
      MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return MIP_OK;
    }();
    (void)hipFree(d_caps);
    if (rc2 != MIP_OK) return rc2;
    ctx->shard_cap_max = mine;
    for (uint32_t c : caps) ctx->shard_cap_max = c > ctx->shard_cap_max ? c : ctx->shard_cap_max;
    const size_t cap = ctx->shard_cap_max;
    size_t stride = (sizeof(MipShardHeader) + cap * 20 + 255) / 256 * 256;  // room for either form of the list
    if (wire_stride_bytes(cap, false) > stride) stride = wire_stride_bytes(cap, false);
    MIP_HIP(ctx, hipMalloc(&ctx->d_send, stride));
    MIP_HIP(ctx, hipMemsetAsync(ctx->d_send, 0, stride, ctx->stream));
    MIP_HIP(ctx, hipMalloc(&ctx->d_recv, stride * world));
    MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MIP_OK;
  }();
  if (rc != MIP_OK) (void)mip_comm_destroy(ctx);  // no half-initialised communicator: the caller may retry with a smaller context
  return rc;
}

int32_t mip_comm_destroy(MipContext* ctx) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (ctx->comm && rccl()) {
    if (int32_t rc = bind_device(ctx)) return rc;
    if (int32_t rc = sync_all(ctx)) return rc;
    (void)rccl()->comm_destroy(ctx->comm);
  }
  ctx->comm = nullptr;
  (void)hipFree(ctx->d_send);
  (void)hipFree(ctx->d_recv);
  ctx->d_send = ctx->d_recv = nullptr;
  return MIP_OK;
}

static int32_t sharded_gather_and_merge(MipContext* ctx, uint32_t cap, void* out_cmds, uint32_t* out_count) {
  if (ctx->sharded_form) {  // the list travels as 8-byte or packed 4-byte records (MIP_OUT_WIRE) and is expanded by the merge
    const bool packed = ctx->sharded_form == 2;
    const uint64_t stride = wire_stride_bytes(cap, packed);
    const ncclResult_t res = rccl()->all_gather(ctx->d_send, ctx->d_recv, stride / 4, ncclUint32, ctx->comm, ctx->stream);
    if (res != ncclSuccess) return fail(ctx, MIP_ERR_DEVICE, "ncclAllGather failed: %s", rccl()->get_error_string ? rccl()->get_error_string(res) : "?");
    ctx->timings.sharded_bytes_sent = stride;
    return enqueue_merge_wire(ctx, ctx->d_recv, ctx->comm_world, stride, cap, out_cmds, out_count, packed);
  }
  const uint64_t stride = (sizeof(MipShardHeader) + (uint64_t)cap * 20 + 255) / 256 * 256;
  ctx->timings.sharded_bytes_sent = stride;
  // ONE all-gather of the fixed-size chunks, then the merge — same stream, no host round trip
  const ncclResult_t res = rccl()->all_gather(ctx->d_send, ctx->d_recv, stride / 4, ncclUint32, ctx->comm, ctx->stream);
  if (res != ncclSuccess) return fail(ctx, MIP_ERR_DEVICE, "ncclAllGather failed: %s", rccl()->get_error_string ? rccl()->get_error_string(res) : "?");
  return enqueue_merge(ctx, ctx->d_recv, ctx->comm_world, stride, cap, out_cmds, out_count);
}

int32_t mip_run_sharded(MipContext* ctx, const MipFrame* frame, const MipShardedOutputs* out) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (!frame || !out || !out->draw_cmds || !out->draw_count) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "frame/out/draw_cmds/draw_count is NULL");
  if (!ctx->comm) return fail(ctx, MIP_ERR_NOT_READY, "mip_comm_init has not been called");
  if (!(out->flags & MIP_OUT_DEVICE)) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "mip_run_sharded needs MIP_OUT_DEVICE");
  const uint32_t cap_max = ctx->shard_cap_max;
  const uint32_t cap = (out->chunk_capacity && out->chunk_capacity < cap_max) ? out->chunk_capacity : cap_max;
  ctx->replay_blocked = true;  // a timed-out sharded frame is reported, never re-issued by one rank on its own (the exchange is collective)
  // 1. this rank's shard, written straight into its chunk (the send buffer always holds max_instances commands)
  MipOutputs local{};
  local.model = out->model;
  local.visible_bitmap = out->visible_bitmap;
  local.world_aabb = out->world_aabb;
  local.draw_count = ctx->d_send;
  local.draw_index_total = ctx->d_send + 1;
  local.draw_cmds = ctx->d_send + sizeof(MipShardHeader) / 4;
  // every rank takes the same form: the mesh table is replicated and shard_cap_max was all-gathered
  int form = ctx->shard_wire;
  if (form == 2 && (uint64_t)cap_max > (1ull << mip_wire_index_bits(ctx->m))) form = 1;
  ctx->sharded_form = form;
  local.flags = MIP_OUT_DEVICE | MIP_OUT_ASYNC | (form ? MIP_OUT_WIRE : 0u) | (form == 2 ? MIP_OUT_WIRE_PACKED : 0u);
  // kernel, all-gather and merge are ordered by ONE stream and share one send/receive buffer: a sharded
  // frame always takes frame slot 0 (= ctx->stream), whatever frames_in_flight is. Overlapping sharded
  // frames is done with several contexts (renderer_amd/sharded.py, PipelinedExchange).
  ctx->next_slot = 0;
  if (int32_t rc = mip_run(ctx, frame, &local)) return rc;
  ctx->next_slot = 0;
  // 2. all-gather, 3. merge
  if (int32_t rc = sharded_gather_and_merge(ctx, cap, out->draw_cmds, out->draw_count)) return rc;
  ctx->sharded_out_cmds = out->draw_cmds;
  ctx->sharded_out_count = out->draw_count;
  ctx->sharded_pending += 1;
  if (out->flags & MIP_OUT_ASYNC) {
    ctx->pending_async = true;
    return MIP_OK;
  }
  return mip_wait(ctx);
}

int32_t mip_import_external_fd(MipContext* ctx, int32_t fd, uint64_t size_bytes, void** out_device_ptr) {
  if (out_device_ptr) *out_device_ptr = nullptr;
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (fd < 0 || size_bytes == 0 || !out_device_ptr) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "bad fd / size / out pointer");
  if (int32_t rc = bind_device(ctx)) return rc;
  hipExternalMemoryHandleDesc hd{};
  hd.type = hipExternalMemoryHandleTypeOpaqueFd;
  hd.handle.fd = fd;
  hd.size = size_bytes;
  hipExternalMemory_t mem = nullptr;
  hipError_t e = hipImportExternalMemory(&mem, &hd);
  if (e != hipSuccess) return fail(ctx, MIP_ERR_DEVICE, "hipImportExternalMemory(OpaqueFd, %llu bytes) failed: %s", (unsigned long long)size_bytes, hipGetErrorString(e));
  hipExternalMemoryBufferDesc bd{};
  bd.offset = 0;
  bd.size = size_bytes;
  void* ptr = nullptr;
  e = hipExternalMemoryGetMappedBuffer(&ptr, mem, &bd);
  if (e != hipSuccess || !ptr) {
    (void)hipDestroyExternalMemory(mem);
    return fail(ctx, MIP_ERR_DEVICE, "hipExternalMemoryGetMappedBuffer failed: %s", hipGetErrorString(e));
  }
  ctx->externals.push_back({mem, ptr});
  *out_device_ptr = ptr;
  return MIP_OK;
}

int32_t mip_release_external(MipContext* ctx, void* device_ptr) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < ctx->externals.size(); ++i)
    if (ctx->externals[i].ptr == device_ptr) {
      if (int32_t rc = bind_device(ctx)) return rc;
      if (int32_t rc = sync_all(ctx)) return rc;
      MIP_HIP(ctx, hipDestroyExternalMemory(ctx->externals[i].mem));
      ctx->externals.erase(ctx->externals.begin() + (long)i);
      return MIP_OK;
    }
  return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "not a pointer returned by mip_import_external_fd");
}

static MipContext::ExternalSemaphore* find_semaphore(MipContext* ctx, MipExternalSemaphore* h, size_t* at = nullptr) {
  for (size_t i = 0; i < ctx->semaphores.size(); ++i)
    if ((void*)ctx->semaphores[i] == (void*)h) {
      if (at) *at = i;
      return ctx->semaphores[i];
    }
  return nullptr;
}

// ---- DRM sync object path (host functions on the stream) ----
struct SemaphoreOp {
  int drm_fd;
  uint32_t handle, kind;
  uint64_t value;
  bool signal;
  volatile uint32_t* error_word;  // host memory (the context's error words): a wait that expired is reported by the next mip_wait
};
constexpr int64_t kSemaphoreWaitNs = 10ll * 1000 * 1000 * 1000;  // bounded like every other wait in the library

// ioctl restarted when a signal interrupts it (what libdrm's drmIoctl does; the waits carry an ABSOLUTE deadline)
static int drm_ioctl(int fd, unsigned long request, void* arg) {
  int rc;
  do rc = ioctl(fd, request, arg);
  while (rc == -1 && (errno == EINTR || errno == EAGAIN));
  return rc;
}

static void semaphore_host_fn(void* p) {
  SemaphoreOp* op = static_cast<SemaphoreOp*>(p);
  uint32_t handle = op->handle;
  uint64_t point = op->value;
  int rc = 0;
  if (op->signal) {
    if (op->kind == MIP_SEMAPHORE_TIMELINE) {
      drm_syncobj_timeline_array a{};
      a.handles = (uintptr_t)&handle;
      a.points = (uintptr_t)&point;
      a.count_handles = 1;
      rc = drm_ioctl(op->drm_fd, DRM_IOCTL_SYNCOBJ_TIMELINE_SIGNAL, &a);
    } else {
      drm_syncobj_array a{};
      a.handles = (uintptr_t)&handle;
      a.count_handles = 1;
      rc = drm_ioctl(op->drm_fd, DRM_IOCTL_SYNCOBJ_SIGNAL, &a);
    }
  } else {
    timespec now;
    clock_gettime(CLOCK_MONOTONIC, &now);
    const int64_t deadline = (int64_t)now.tv_sec * 1000000000ll + now.tv_nsec + kSemaphoreWaitNs;
    if (op->kind == MIP_SEMAPHORE_TIMELINE) {
      drm_syncobj_timeline_wait w{};
      w.handles = (uintptr_t)&handle;
      w.points = (uintptr_t)&point;
      w.timeout_nsec = deadline;
      w.count_handles = 1;
      w.flags = DRM_SYNCOBJ_WAIT_FLAGS_WAIT_FOR_SUBMIT;  // the point may not have been submitted yet
      rc = drm_ioctl(op->drm_fd, DRM_IOCTL_SYNCOBJ_TIMELINE_WAIT, &w);
    } else {
      drm_syncobj_wait w{};
      w.handles = (uintptr_t)&handle;
      w.timeout_nsec = deadline;
      w.count_handles = 1;
      w.flags = DRM_SYNCOBJ_WAIT_FLAGS_WAIT_FOR_SUBMIT;
      rc = drm_ioctl(op->drm_fd, DRM_IOCTL_SYNCOBJ_WAIT, &w);
      if (rc == 0) {  // a binary semaphore is consumed by its wait
        drm_syncobj_array a{};
        a.handles = (uintptr_t)&handle;
        a.count_handles = 1;
        (void)drm_ioctl(op->drm_fd, DRM_IOCTL_SYNCOBJ_RESET, &a);
      }
    }
  }
  if (rc != 0) *op->error_word = mip::kErrSemaphore;
  delete op;
}

static int32_t enqueue_semaphore_op(MipContext* ctx, MipContext::ExternalSemaphore* s, uint64_t value, bool signal, hipStream_t stream) {
  SemaphoreOp* op = new (std::nothrow) SemaphoreOp{ctx->drm_fd, s->drm_handle, s->kind, value, signal, ctx->h_error + 5};
  if (!op) return fail(ctx, MIP_ERR_OUT_OF_MEMORY, "out of host memory");
  const hipError_t e = hipLaunchHostFunc(stream, semaphore_host_fn, op);
  if (e != hipSuccess) {
    delete op;
    return fail(ctx, MIP_ERR_DEVICE, "hipLaunchHostFunc failed: %s", hipGetErrorString(e));
  }
  return MIP_OK;
}

int32_t mip_import_external_semaphore_fd(MipContext* ctx, int32_t fd, uint32_t kind, MipExternalSemaphore** out_semaphore) {
  if (out_semaphore) *out_semaphore = nullptr;
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  if (fd < 0 || !out_semaphore) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "bad fd / out pointer");
  if (kind != MIP_SEMAPHORE_BINARY && kind != MIP_SEMAPHORE_TIMELINE) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "kind %u is neither MIP_SEMAPHORE_BINARY nor MIP_SEMAPHORE_TIMELINE", kind);
  if (int32_t rc = bind_device(ctx)) return rc;
  hipExternalSemaphoreHandleDesc hd{};
  hd.type = kind == MIP_SEMAPHORE_TIMELINE ? hipExternalSemaphoreHandleTypeTimelineSemaphoreFd : hipExternalSemaphoreHandleTypeOpaqueFd;
  hd.handle.fd = fd;
  hipExternalSemaphore_t sem = nullptr;
  hipError_t e = std::getenv("MIP_TUNE_SEMAPHORE_VIA_DRM") ? hipErrorNotSupported : hipImportExternalSemaphore(&sem, &hd);
  uint32_t drm_handle = 0;
  if (e != hipSuccess || !sem) {
    // The runtime refuses the handle type (ROCm 7.2, Linux: TimelineSemaphoreFd -> "invalid argument", OpaqueFd ->
    // "operation not supported"). The fd itself is a kernel sync object: take it on a render node.
    sem = nullptr;
    (void)hipGetLastError();
    if (ctx->drm_fd < 0) {
      char node[64];
      for (int k = 128; k < 192 && ctx->drm_fd < 0; ++k) {
        snprintf(node, sizeof node, "/dev/dri/renderD%d", k);
        ctx->drm_fd = open(node, O_RDWR | O_CLOEXEC);
      }
    }
    drm_syncobj_handle h{};
    h.fd = fd;
    if (ctx->drm_fd < 0 || drm_ioctl(ctx->drm_fd, DRM_IOCTL_SYNCOBJ_FD_TO_HANDLE, &h) != 0 || !h.handle)
      return fail(ctx, MIP_ERR_DEVICE, "hipImportExternalSemaphore(%s) failed: %s; and the fd is not a DRM sync object either (%s)",
                  kind == MIP_SEMAPHORE_TIMELINE ? "TimelineSemaphoreFd" : "OpaqueFd", hipGetErrorString(e),
                  ctx->drm_fd < 0 ? "no render node could be opened" : "DRM_IOCTL_SYNCOBJ_FD_TO_HANDLE refused it");
    drm_handle = h.handle;
    close(fd);  // imported: the fd belonged to the library from here on (the sync object lives on through the handle)
  }
  auto* entry = new (std::nothrow) MipContext::ExternalSemaphore{sem, kind, drm_handle};
  if (!entry) {
    if (sem) (void)hipDestroyExternalSemaphore(sem);
    return fail(ctx, MIP_ERR_OUT_OF_MEMORY, "out of host memory");
  }
  ctx->semaphores.push_back(entry);
  *out_semaphore = (MipExternalSemaphore*)entry;
  return MIP_OK;
}

int32_t mip_external_semaphore_on_device(MipContext* ctx, MipExternalSemaphore* semaphore) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  MipContext::ExternalSemaphore* s = find_semaphore(ctx, semaphore);
  if (!s) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "not a semaphore returned by mip_import_external_semaphore_fd");
  return s->sem ? 1 : 0;
}

int32_t mip_wait_external(MipContext* ctx, MipExternalSemaphore* semaphore, uint64_t value) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  MipContext::ExternalSemaphore* s = find_semaphore(ctx, semaphore);
  if (!s) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "not a semaphore returned by mip_import_external_semaphore_fd");
  if (int32_t rc = bind_device(ctx)) return rc;
  ctx->replay_blocked = true;  // with a consumer ordered by semaphores a frame is never silently issued twice
  // the stream the NEXT frame will be enqueued on: that frame then starts only when the semaphore has been reached
  hipStream_t stream = ctx->slots[ctx->next_slot].stream;
  if (s->sem) {
    hipExternalSemaphoreWaitParams p{};
    p.params.fence.value = value;
    MIP_HIP(ctx, hipWaitExternalSemaphoresAsync(&s->sem, &p, 1, stream));
  } else if (int32_t rc = enqueue_semaphore_op(ctx, s, value, false, stream)) {
    return rc;
  }
  ctx->pending_async = true;
  return MIP_OK;
}

int32_t mip_signal_external(MipContext* ctx, MipExternalSemaphore* semaphore, uint64_t value) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  MipContext::ExternalSemaphore* s = find_semaphore(ctx, semaphore);
  if (!s) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "not a semaphore returned by mip_import_external_semaphore_fd");
  if (int32_t rc = bind_device(ctx)) return rc;
  ctx->replay_blocked = true;
  // behind the frame that was issued last (its slot's stream)
  hipStream_t stream = ctx->slots[ctx->last_slot].stream;
  if (s->sem) {
    hipExternalSemaphoreSignalParams p{};
    p.params.fence.value = value;
    MIP_HIP(ctx, hipSignalExternalSemaphoresAsync(&s->sem, &p, 1, stream));
  } else if (int32_t rc = enqueue_semaphore_op(ctx, s, value, true, stream)) {
    return rc;
  }
  ctx->pending_async = true;
  return MIP_OK;
}

int32_t mip_release_external_semaphore(MipContext* ctx, MipExternalSemaphore* semaphore) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
  size_t at = 0;
  MipContext::ExternalSemaphore* s = find_semaphore(ctx, semaphore, &at);
  if (!s) return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "not a semaphore returned by mip_import_external_semaphore_fd");
  if (int32_t rc = bind_device(ctx)) return rc;
  if (int32_t rc = sync_all(ctx)) return rc;
  if (s->sem) MIP_HIP(ctx, hipDestroyExternalSemaphore(s->sem));
  if (s->drm_handle) {
    drm_syncobj_destroy d{};
    d.handle = s->drm_handle;
    (void)ioctl(ctx->drm_fd, DRM_IOCTL_SYNCOBJ_DESTROY, &d);
  }
  ctx->semaphores.erase(ctx->semaphores.begin() + (long)at);
  delete s;
  return MIP_OK;
}

}  // extern "C"

namespace {
// A merge found a chunk whose header count exceeds the exchanged capacity (a tightened chunk and a
// camera that moved). Every rank sees the same gathered headers, so every rank gets here for the same
// frame: the all-gather + merge of THAT frame are repeated once at full capacity — this rank's
// complete list is still in the send buffer — unless a later sharded frame has overwritten it.
int32_t repair_sharded_overflow(MipContext* ctx) {
  if (!ctx->comm || !ctx->sharded_out_cmds || ctx->sharded_pending != 1) {
    ctx->sharded_pending = 0;
    return fail(ctx, MIP_ERR_CAPACITY,
                "a shard's draw list is longer than the exchanged chunk holds; merged list truncated%s",
                ctx->comm ? " (more than one sharded frame was in flight: the overflowing one can no longer be re-sent)" : "");
  }
  ctx->sharded_pending = 0;
  const uint32_t cap_max = ctx->shard_cap_max;
  if (int32_t rc = sharded_gather_and_merge(ctx, cap_max, ctx->sharded_out_cmds, ctx->sharded_out_count)) return rc;
  MIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->timings.sharded_retries += 1;
  uint32_t e = 0;
  for (uint32_t k = 0; k < mip::kErrWords; ++k) {
    if (k == 0 || k == 4) continue;  // a local timeout of the frame itself is reported by the caller (check_device_error)
    e |= ((volatile uint32_t*)ctx->h_error)[k];
    ((volatile uint32_t*)ctx->h_error)[k] = 0;
  }
  if (e) return fail(ctx, MIP_ERR_DEVICE, "sharded repair failed (device error bits %u)", e);
  return MIP_OK;
}
}  // namespace

extern "C" {

const char* mip_last_error(const MipContext* ctx) { return ctx ? ctx->err : "null context"; }

int32_t mip_get_timings(MipContext* ctx, MipTimings* out) {
  if (!ctx || !out) return MIP_ERR_INVALID_ARGUMENT;
  *out = ctx->timings;
  return MIP_OK;
}

int32_t mip_reset_timings(MipContext* ctx) {
  if (!ctx) return MIP_ERR_INVALID_ARGUMENT;
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
