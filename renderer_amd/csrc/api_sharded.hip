// api_sharded.hip — C ABI of the instance pipeline, part 3 of 4: scenes sharded over several GPUs.
// The merge kernels of the all-gathered shard lists, the collective-library seam (RCCL by default, opened with dlopen),
// mip_run_sharded = shard kernel -> ONE ncclAllGather -> merge on one stream, and the collective repair of a frame
// whose tightened chunk overflowed.
#include "context.hpp"
#include "merge_kernel.hpp"  // row e: instantiated here and only here

#include <dlfcn.h>

namespace mip_host {
namespace {

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

}  // namespace

void comm_release(MipContext* ctx) {
  if (ctx->comm && rccl()) (void)rccl()->comm_destroy(ctx->comm);
  ctx->comm = nullptr;
  (void)hipFree(ctx->d_send);
  (void)hipFree(ctx->d_recv);
  ctx->d_send = ctx->d_recv = nullptr;
}

}  // namespace mip_host

using namespace mip_host;

extern "C" {

static int32_t enqueue_merge(MipContext* ctx, const void* chunks, uint32_t n_chunks, uint64_t chunk_stride_bytes,
                             uint32_t chunk_capacity, void* out_cmds, uint32_t* out_count) {
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
                MIP_WIRE_BLOCK_HEADER_BYTES == mip::kWireBlockHeaderWords * 4u && MIP_WIRE_SUB_BLOCK_COMMANDS == mip::kWireSubBlock &&
                MIP_WIRE_PACKED_BLOCK_COMMANDS == mip::kWirePackedBlockCmds && MIP_WIRE_PACKED_BLOCK_BYTES == mip::kWirePackedBlockWords * 4u,
                "wire layout: header and kernels agree");
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
  // one WAVE expands one group of 256 records at a time (four waves per workgroup); sized for the groups the chunks can hold
  const uint64_t max_groups = ((uint64_t)chunk_capacity + mip::kMergeGroupCmds - 1) / mip::kMergeGroupCmds * n_chunks;
  const uint64_t max_blocks = (max_groups + 3) / 4;
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
  if ((uintptr_t)chunks & 15u)  // block headers and records are read with 16-byte loads
    return fail(ctx, MIP_ERR_INVALID_ARGUMENT, "wire chunks must be 16-byte aligned");
  if (chunk_capacity == 0) {  // what the stride holds, in whole blocks
    const uint64_t fits = packed ? (chunk_stride_bytes - sizeof(MipShardHeader)) / MIP_WIRE_PACKED_BLOCK_BYTES * MIP_WIRE_PACKED_BLOCK_COMMANDS
                                 : (chunk_stride_bytes - sizeof(MipShardHeader)) / MIP_WIRE_BLOCK_BYTES * MIP_WIRE_BLOCK_COMMANDS;
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

}  // extern "C"

namespace mip_host {
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
    e |= ((volatile uint32_t*)ctx->h_error)[k];
    ((volatile uint32_t*)ctx->h_error)[k] = 0;
  }
  if (e) return fail(ctx, MIP_ERR_DEVICE, "sharded repair failed (device error bits %u)", e);
  return MIP_OK;
}
}  // namespace mip_host
