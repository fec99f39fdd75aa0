// context.hpp — the state behind a MipContext and the helpers the four host translation units share.
//   api_context.hip   create / destroy, uploads (mesh table, instances, geometry, skeleton, poses), census, diagnostics
//   api_frame.hip     one frame: plan (frame_plan.hpp) -> launches; recorded launch graphs; views; light lists; mip_wait
//   api_sharded.hip   shard merges, the collective-library seam, mip_run_sharded and its collective repair
//   api_interop.hip   external memory and external semaphores (row f-2)
// (round 3 had all of it in one 2 224-line mip_api.hip)
#pragma once

#include "../../include/mi_instance_pipeline.h"
#include "frame_plan.hpp"
#include "instance_kernel.hpp"  // rows a-1 .. a-7 (the kernel itself is instantiated in api_frame.hip and stages_tu.hip only)
#include "stage_args.hpp"      // argument blocks + launchers of everything built in stages_tu.hip

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is opened with dlopen, never linked

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

constexpr uint32_t kHelpWords = mip::kHelpShards + 1;  // MipContext::d_help
constexpr uint32_t kHelpHintWord = 8;                  // of MipContext::h_error (the error words are [0, mip::kErrWords))
constexpr uint32_t kFirstMoverLaunches = 8;            // launches that follow the first-mover rule after a hint
static_assert(kHelpHintWord >= mip::kErrWords, "the hint must not be an error word");

struct MipContext {
  int device = -1;
  uint32_t max_instances = 0, max_meshes = 0, cfg_flags = 0;
  uint32_t n = 0, m = 0;
  bool have_instances = false, have_meshes = false;
  int force_order = 0;         // tuning (MIP_TUNE_ORDER): 1 or 3, 0 = by instance count
  bool force_general = false;  // tuning/tests (MIP_TUNE_FORCE_GENERAL): always launch the kernel with the literal cold path
  // One slot per frame in flight: its own stream and its own cross-tile prefix state, so that
  // consecutive frames may overlap on the device (MipConfig.frames_in_flight).
  struct FrameSlot {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    unsigned long long* d_status = nullptr;  // level-0 granules, accumulators, group starts
    uint32_t* d_scalars = nullptr;           // [0] draw_count, [1] index_total (host-output runs), [2] pre-triangle count, [3] command ticket
    uint32_t* d_tmp_cmds = nullptr;          // per-triangle stage: the instance kernel's list before re-compaction
    uint32_t* d_tmp_src = nullptr;           //                     and each command's source index offset
    unsigned long long* d_tmp_blocks = nullptr;  //                 re-compaction of large frames: one granule per 1024 commands
    uint32_t recompact_epoch = 0;            //                     tag of the last one-launch re-compaction on this slot
    bool tri_ticket_clean = false;           //                     the stage's ticket word (d_scalars + 3) was cleared by the last frame's re-compaction
    bool tri_sort_clean = false;             //                     ... and so was d_tri_sort
    uint32_t* d_tmp_final = nullptr;         //                     parts kernel: each command's final indexCount (never written into the command)
    unsigned long long* d_part_status = nullptr;  //                small frames: one granule per (command, part)
    uint32_t tri_epoch = 0;                  //                     tag of the last parts launch on this slot
    uint32_t* d_chunk_first = nullptr;       // range kernel (round 5): first command of every range of the triangle stream
    unsigned long long* d_chunk_status = nullptr;  //                   one granule per range
    size_t chunks_cap = 0;                   //                         ranges both arrays hold
    uint32_t chunk_epoch = 0;                //                         tag of the last range launch on this slot
    uint32_t* d_tri_order = nullptr;         // large frames: command numbers by descending size class
    uint32_t* d_tri_sort = nullptr;          //               kSortWords words: histogram, positions, tickets (triangle_kernels.hpp)
    float* d_skin_box = nullptr;             // skinned frames: per instance posed mesh-space box {min xyz, -, max xyz, -}
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
  uint32_t acc1_offset_words = 0, start1_offset_words = 0, helps_seen_offset_words = 0, groups_cap = 0;
  // the frame kernel's first-mover rule (instance_kernel.hpp, KernelArgs.first_mover_rule): followed by the launches that come
  // after a launch whose tile 0 saw new helps (it writes the count to h_error[kHelpHintWord])
  bool no_one_mesh = false;               // MIP_TUNE_NO_ONE_MESH: a one-entry mesh table is gathered from like any other (A/B runs, tests of that path)
  uint32_t first_mover_env = 0;           // MIP_TUNE_FIRST_MOVER=always|never -> 1|2, read at context creation; 0: as the hints say
  uint32_t help_hint_seen = 0;            // the hint word's value when the host last looked
  uint32_t first_mover_launches_left = 0; // launches that still follow the rule
  uint32_t lds_pad = 0;  // tuning only (MIP_TUNE_LDS_PAD): dynamic LDS bytes that cap workgroups per CU
  uint32_t tri_block_threads = 0;  // tuning (MIP_TUNE_TRI_BLOCK_THREADS): 256 / 512 / 1024, 0 = by instance count
  uint32_t tri_block_max = 65536;  // instance counts up to this use the workgroup-per-command triangle kernel
  uint32_t tri_parts_max = 1024;   // instance counts up to this use the parts kernel (16 work items per command), 0 = off
  bool tri_recompact_three_launches = false;  // MIP_TUNE_TRI_RECOMPACT_LAUNCHES=3: round 4's count / scan / scatter instead of the one-launch form (A/B, tests)
  uint32_t tri_ticket_slots = 4096; // long triangle streams: slots per range of the range kernel (MIP_TUNE_TRI_RANGE_SLOTS: 256 .. 8192, whole steps of 64)
  uint32_t tri_chunks_from = 0;    // instance counts from this use the range kernel; MIP_TUNE_TRI_CHUNKS_FROM (4294967295 = never: the round-4 kernels)
  bool tri_no_choice = false;      // tuning (MIP_TUNE_TRI_NO_CHOICE): large frames always take the wave-per-command kernel
  int tri_force_choice = 0;        // tuning (MIP_TUNE_TRI_CHOICE=block|waves): 1 / 2 force the device-side choice of the large-frame grid
  uint32_t tri_batch_from = 65536; // commands from which the ticket-pulling workgroup kernel takes four per ticket (MIP_TUNE_TRI_BATCH_FROM: tests)
                                   // measured (DamagedHelmet entry, frame time parts / workgroup-per-command): 30 instances 14 / 24 us,
                                   // 200: 16 / 25, 1000: 41 / 47, 2000: 67 / 64, 4000: 113 / 83
  uint32_t max_lod_tris = 0;       // largest triangle count of LOD 0 / LOD 1 over the mesh table
  // upload-time census of instances that fail the kernel's finite test (instance_kernel.hpp,
  // finite_magnitude): while it is zero, frames run the kernel without the literal cold path
  uint64_t nonfinite_instances = 0;
  float box_abs = 0.f;  // largest sum of |box coordinates| over the mesh table (the census' overflow bound)
  uint32_t* d_census = nullptr;
  uint32_t* h_error = nullptr;  // pinned, device-visible: error words [0, kErrWords)
  uint32_t* d_help = nullptr;   // device memory, kHelpWords words: helped tile aggregates (MipTimings.prefix_helps = their sum); read back by mip_get_timings
  uint32_t help_base = 0;       // its value at the last mip_reset_timings
  uint32_t* d_error = nullptr;  // device alias of h_error
  uint32_t carried_error_bits = 0;  // error bits a synchronous call saw while asynchronous work was in flight: reported then AND by the next mip_wait
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
  struct ExternalSemaphore {
    hipExternalSemaphore_t sem;
    uint32_t kind;
    uint32_t drm_handle;
    // DRM path, stream-value hand-over (api_interop.hip): two pinned, device-visible sequence words —
    // [0] waits granted by the waiter thread (the stream holds a hipStreamWaitValue64 on it),
    // [1] signals reached by the stream (hipStreamWriteValue64; the signaller thread polls it) — and the sequence numbers issued so far
    unsigned long long* words;
    unsigned long long wait_seq, signal_seq;
  };
  std::vector<ExternalSemaphore*> semaphores;
  struct SemaphoreWorkers;             // the two helper threads and their queues (api_interop.hip); created with the first DRM-path semaphore
  SemaphoreWorkers* semaphore_workers = nullptr;
  int drm_fd = -1;  // render node, opened on first use
  uint32_t last_slot = 0;  // slot of the frame issued last (mip_signal_external goes behind it)
  char err[512] = {0};
#ifdef MIP_DEBUG_STAMPS
  unsigned long long* d_stamps = nullptr;
#endif
};

namespace mip_host {

int32_t fail(MipContext* ctx, int32_t code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));

#define MIP_HIP(ctx, call)                                                                    \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return mip_host::fail(ctx, e_ == hipErrorOutOfMemory ? MIP_ERR_OUT_OF_MEMORY : MIP_ERR_DEVICE, \
                            "%s failed: %s", #call, hipGetErrorString(e_));                   \
  } while (0)

inline uint32_t tiles_for(uint32_t n) { return (n + mip::kTile - 1) / mip::kTile; }
static_assert(mip::kPlanTile == mip::kTile && mip::kPlanTriParts == mip::kTriParts && mip::kPlanTriPartMaxT == mip::kTriPartMaxT,
              "frame_plan.hpp restates kernel constants");

int32_t bind_device(MipContext* ctx);
int32_t sync_all(MipContext* ctx);
// Reads and clears the device-visible error words after streams have drained; runs the collective repair of a sharded
// frame when the merge kernel asked for it (api_sharded.hip).
int32_t check_device_error(MipContext* ctx);
// Number of instances of [first, first + count) that fail the finite test, and (bad_ids != null) how many name a mesh
// outside a table of `m` entries. Synchronous.
int32_t census(MipContext* ctx, uint32_t first, uint32_t count, uint32_t* out, uint32_t* bad_ids = nullptr, uint32_t m = 0);
void drop_graphs(MipContext* ctx);                    // api_frame.hip
int32_t repair_sharded_overflow(MipContext* ctx);     // api_sharded.hip
void comm_release(MipContext* ctx);                   // api_sharded.hip: communicator + buffers, for mip_destroy
void interop_release(MipContext* ctx);                // api_interop.hip: imported memory and semaphores, for mip_destroy
int32_t interop_drain(MipContext* ctx);                // api_interop.hip: every queued signal of an external semaphore has been performed (mip_wait)
int32_t run_frame(MipContext* ctx, const MipFrame* frame, const MipOutputs* out, bool skinned, void* palette);  // api_frame.hip

}  // namespace mip_host
